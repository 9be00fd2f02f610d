#!/bin/bash
# As build_variant.sh, for hg_insert.hip: scripts/libhg_<name>.so = the tree's library with hg_insert.hip compiled with
# extra flags (timing experiments; HG_LIB_PATH selects it).  Usage: bash scripts/build_variant_insert.sh name "-D..."
set -e
name=$1; extra=$2
root=$(cd $(dirname $0)/.. && pwd)
src=$root/hectorgrapher_amd/csrc
tmp=/tmp/hg_variant_$name; mkdir -p $tmp
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-result -I$src $extra"
/opt/rocm/bin/hipcc $F -c $src/hg_insert.hip -o $tmp/hg_insert.o
(cd $src && make -s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$src/hg_exports.map -o $root/scripts/libhg_$name.so \
  $src/hg_grid.o $tmp/hg_insert.o $src/hg_filter.o $src/hg_xray.o $src/hg_unwarp.o $src/hg_match.o $src/hg_match_big.o
echo built scripts/libhg_$name.so
