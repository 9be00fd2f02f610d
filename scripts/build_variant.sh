#!/bin/bash
# Builds scripts/libhg_<name>.so: the library with hg_match.hip (both solver builds) compiled with extra flags, the
# other objects taken from the tree's build. For A/B runs and diagnostic builds (HG_LIB_PATH=... selects it).
# Usage: bash scripts/build_variant.sh name "-DHG_LM_STAMPS ..."
set -e
name=$1; extra=$2
root=$(cd $(dirname $0)/.. && pwd)
src=$root/hectorgrapher_amd/csrc
tmp=/tmp/hg_variant_$name; mkdir -p $tmp
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-result -I$src $extra"
/opt/rocm/bin/hipcc $F -c $src/hg_match.hip -o $tmp/hg_match.o &
/opt/rocm/bin/hipcc $F -DHG_BIG -c $src/hg_match.hip -o $tmp/hg_match_big.o &
wait
(cd $src && make -s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$src/hg_exports.map -o $root/scripts/libhg_$name.so \
  $src/hg_grid.o $src/hg_insert.o $src/hg_filter.o $src/hg_xray.o $src/hg_unwarp.o $tmp/hg_match.o $tmp/hg_match_big.o
echo built scripts/libhg_$name.so
