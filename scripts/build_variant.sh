#!/bin/bash
# Builds an alternative libfwgpu.so with extra -D flags for kernels.hip (kernel tuning A/B runs; select it with FWGPU_LIBRARY).
# usage: scripts/build_variant.sh NAME -DFW_UO=4 ...   ->  build/variants/libfwgpu_NAME.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $R/build/variants
cd $R/fwumious_wabbit_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I$R/include -I. -mllvm -pragma-unroll-threshold=131072 "$@" -c kernels.hip -o /tmp/k_$NAME.o
O=$R/fwumious_wabbit_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libfwgpu_$NAME.so /tmp/k_$NAME.o $(ls $O/*.o | grep -v "/kernels.o") -lz -ldl
echo built $R/build/variants/libfwgpu_$NAME.so
