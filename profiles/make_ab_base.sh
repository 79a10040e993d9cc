#!/bin/bash
# A baseline for A/B runs ACROSS source revisions: LD_PRELOADing another revision's library under this revision's host is
# not valid once kernel signatures differ (both libraries register their kernels under the same host stubs), so the
# baseline is a complete pair -- that revision's library and its own vcm_render -- built in /tmp and copied to
# profiles/ab_base/{csrc,host}/ (untracked; travels to the GPU box).  quick_ab.sh runs it as the row "base" with BASE=1.
#   bash profiles/make_ab_base.sh [git-rev]        (default HEAD: the last commit, against the working tree's changes)
set -e
REV=${1:-HEAD}
cd "$(dirname "$0")/.."
ROOT=$PWD
T=/tmp/ab_base_src; rm -rf $T; mkdir -p $T
git archive $REV smallvcm_amd/csrc smallvcm_amd/host include | tar -x -C $T
make -C $T/smallvcm_amd/csrc > /tmp/ab_base_build.log 2>&1
make -C $T/smallvcm_amd/host vcm_render >> /tmp/ab_base_build.log 2>&1
mkdir -p $ROOT/profiles/ab_base/csrc $ROOT/profiles/ab_base/host
cp $T/smallvcm_amd/csrc/libsmallvcm_amd.so $ROOT/profiles/ab_base/csrc/
cp $T/smallvcm_amd/host/vcm_render $ROOT/profiles/ab_base/host/
git rev-parse --short $REV > $ROOT/profiles/ab_base/REV
echo "baseline $(cat $ROOT/profiles/ab_base/REV) in profiles/ab_base"
