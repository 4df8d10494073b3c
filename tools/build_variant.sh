#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: libhibag_hip.so with extra compiler flags on the kernels and the host code that
# shares their headers, as gpurun_var_NAME.so at the repo root (tools/run_variants.sh times each of them on the GPU box).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=hibag_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-result"
# (the kernels' extra LLVM option comes from the Makefile's probe; NO_KFLAGS=1 leaves it out: the other half of the parity check)
KFLAGS=$([ -n "${NO_KFLAGS:-}" ] || make -s -C $src print-kflags)
/opt/rocm/bin/hipcc $FLAGS $KFLAGS "$@" -c $src/hibag_kernels.hip -o /tmp/var_$name.o
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src/hibag_model.hip -o /tmp/var_${name}_model.o
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src/hibag_predict.hip -o /tmp/var_${name}_predict.o
# (every other object as the Makefile built it: the list is the Makefile's own)
others=$(ls $src/*.o | grep -v -e hibag_kernels.o -e hibag_model.o -e hibag_predict.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_var_$name.so /tmp/var_$name.o /tmp/var_${name}_model.o /tmp/var_${name}_predict.o $others -ldl
echo built gpurun_var_$name.so
