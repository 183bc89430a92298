#!/bin/bash
# builds the product library with the round-3 inflate kernel (tools/experiments/inflate_r3.hip) as tools/experiments/libclairsto_r3inflate.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/clairs_to_amd/csrc
make -s -j8
mkdir -p build_r3
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I. -c $R/tools/experiments/inflate_r3.hip -o build_r3/inflate.o
OBJS=$(ls build/*.o | grep -v "build/inflate.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/experiments/libclairsto_r3inflate.so $OBJS build_r3/inflate.o -lpthread -lz -ldl
ls -la $R/tools/experiments/libclairsto_r3inflate.so
