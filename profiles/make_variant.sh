#!/bin/bash
# A measurement variant of the library as a complete pair for A/B runs: profiles/ab_<name>/{csrc/libsmallvcm_amd.so,
# host/vcm_render} (untracked; vcm_render finds its library through RUNPATH $ORIGIN/../csrc).  LD_PRELOADing a variant under
# the default host is NOT valid: both libraries register their kernels under the same host stubs, and the kernels that run
# are then not the variant's (round 4: a variant with another buffer layout produced NaNs that way).
#   bash profiles/make_variant.sh <name> "<extra compiler flags>"
set -e
NAME=$1; EXTRA=$2
cd "$(dirname "$0")/.."
make -C smallvcm_amd/csrc variant NAME=$NAME EXTRA="$EXTRA" 2>&1 | grep -i "error\|warning" || true
make -C smallvcm_amd/host vcm_render > /dev/null
mkdir -p profiles/ab_$NAME/csrc profiles/ab_$NAME/host
mv smallvcm_amd/csrc/libsmallvcm_amd_$NAME.so profiles/ab_$NAME/csrc/libsmallvcm_amd.so
cp smallvcm_amd/host/vcm_render profiles/ab_$NAME/host/
echo "$NAME: $EXTRA" > profiles/ab_$NAME/REV
echo "variant $NAME in profiles/ab_$NAME"
