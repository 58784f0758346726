#!/bin/bash
# Build libnrhip.so (HIP, gfx950) in-tree and the oracle's C restatement.  hipcc cross-compiles without a GPU.
# The translation units are compiled in parallel into nuradiomc_amd/lib/obj/ and linked; extra arguments go to every hipcc
# compile (e.g. ./build.sh -DATT_WAVES=3).  NRHIP_LIB_NAME=libnrhip_x.so builds a variant next to the default library.
set -e
cd "$(dirname "$0")"
mkdir -p nuradiomc_amd/lib/obj oracle/_build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=nuradiomc_amd/lib/${NRHIP_LIB_NAME:-libnrhip.so}
TAG=$(echo "${NRHIP_LIB_NAME:-libnrhip.so}" | tr -c 'A-Za-z0-9' '_')
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I include -I nuradiomc_amd/csrc"
OBJS=""
PIDS=""
# The ray tracer and the attenuation quadrature are bit-equal to the CPU checker: nothing may be fused there but the explicit
# fma() calls (-ffp-contract=off).  The spectral chain, the ARZ model and the birefringent propagation are held to 1e-6 / 1e-9
# relative, not to bits: there a * b + c may become one v_fma_f64 (a third fewer FP64 instructions in the complex arithmetic).
for f in api raytrace raytrace_refl arz birefringence earth attenuation comm cull spectral pipeline; do
    o=nuradiomc_amd/lib/obj/${TAG}_$f.o
    OBJS="$OBJS $o"
    CONTRACT=""
    case $f in spectral|arz|birefringence) CONTRACT="-ffp-contract=${NRHIP_SPECTRAL_CONTRACT:-fast-honor-pragmas}";; esac
    $HIPCC $FLAGS $CONTRACT "$@" -c nuradiomc_amd/csrc/$f.hip -o $o &
    PIDS="$PIDS $!"
done
for p in $PIDS; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT -Wl,-rpath,/opt/rocm/lib -Wl,-z,defs -ldl   # (-z defs: a declaration without its definition fails here, not at dlopen)
gcc -O2 -fPIC -shared -std=gnu11 -ffp-contract=off -o oracle/_build/liboracle.so oracle/nrmc_oracle.c oracle/arz_oracle.c -lm
echo "built $OUT oracle/_build/liboracle.so"
