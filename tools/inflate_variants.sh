#!/bin/bash
# variant builds of the inflate kernel next to the product library: tools/experiments/libclairsto_inf_<tag>.so
#   bash tools/inflate_variants.sh tag1 "-DCTO_INF_TBL=9 ..." tag2 "..." ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/clairs_to_amd/csrc
make -s -j8
OBJS=$(ls build/*.o | grep -v "build/inflate.o")
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  mkdir -p build_inf_$tag
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c inflate.hip -o build_inf_$tag/inflate.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/experiments/libclairsto_inf_$tag.so $OBJS build_inf_$tag/inflate.o -lpthread -lz -ldl
  echo built $tag "$flags"
done
