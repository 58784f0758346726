#!/bin/bash
# Build libnrhip.so (HIP, gfx950) in-tree and the oracle's C restatement.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
mkdir -p nuradiomc_amd/lib oracle/_build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRC="nuradiomc_amd/csrc/api.hip nuradiomc_amd/csrc/raytrace.hip nuradiomc_amd/csrc/raytrace_refl.hip nuradiomc_amd/csrc/arz.hip nuradiomc_amd/csrc/birefringence.hip nuradiomc_amd/csrc/earth.hip nuradiomc_amd/csrc/attenuation.hip nuradiomc_amd/csrc/comm.hip"
[ -f nuradiomc_amd/csrc/spectral.hip ] && SRC="$SRC nuradiomc_amd/csrc/spectral.hip"
[ -f nuradiomc_amd/csrc/pipeline.hip ] && SRC="$SRC nuradiomc_amd/csrc/pipeline.hip"
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function \
    -I include -I nuradiomc_amd/csrc $SRC -o nuradiomc_amd/lib/libnrhip.so -Wl,-rpath,/opt/rocm/lib -ldl "$@"
gcc -O2 -fPIC -shared -std=gnu11 -ffp-contract=off -o oracle/_build/liboracle.so oracle/nrmc_oracle.c oracle/arz_oracle.c -lm
echo "built nuradiomc_amd/lib/libnrhip.so oracle/_build/liboracle.so"
