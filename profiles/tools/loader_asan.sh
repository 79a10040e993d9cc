#!/bin/bash
# The scene loader (plain C++: smallvcm_amd/csrc/scene_file.cpp + scene_cornell.cpp) under AddressSanitizer + UBSan against 6000
# mangled files.  Round 4: no report (profiles/archive/r06_loader_asan.txt).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p /tmp/asan
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off -I$ROOT/include \
    -o /tmp/asan/libscene_asan.so $ROOT/smallvcm_amd/csrc/scene_file.cpp $ROOT/smallvcm_amd/csrc/scene_cornell.cpp
for seed in 7 99; do
  ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    SCENE_LIB=/tmp/asan/libscene_asan.so python3 $ROOT/profiles/tools/loader_fuzz.py $seed
done
