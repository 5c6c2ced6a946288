#!/bin/bash
# development aid: scripts/dev/libdun_hip_exp.so = the library with gemm.hip compiled with the given -D flags
# (A/B builds of GEMM experiments; use with DUN_HIP_LIB=scripts/dev/libdun_hip_exp.so)
set -e
cd "$(dirname "$0")/../.."
O=dun_amd/build/exp; mkdir -p $O
for f in api elbo head trunk smallnet persist; do cp dun_amd/build/$f.o $O/$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c dun_amd/csrc/gemm.hip -o $O/gemm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/dev/libdun_hip_exp.so $O/*.o
echo scripts/dev/libdun_hip_exp.so
