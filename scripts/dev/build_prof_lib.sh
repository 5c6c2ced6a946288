#!/bin/bash
# development aid: libdun_hip_prof.so = the library with -DPN_PROFILE (phase stamps of the one-launch small-net kernels)
set -e
cd "$(dirname "$0")/../.."
O=dun_amd/build/prof; mkdir -p $O
for f in api gemm elbo head trunk smallnet; do cp dun_amd/build/$f.o $O/$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DPN_PROFILE -c dun_amd/csrc/persist.hip -o $O/persist.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/dev/libdun_hip_prof.so $O/*.o
echo scripts/dev/libdun_hip_prof.so
