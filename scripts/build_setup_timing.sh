#!/bin/bash
# Development build that prints the host time of every step of the window analysis (HIPBA_SETUP_TIMING) to stderr
# -> scripts/dev/libhipba_setup.so (the other objects come from ./build.sh)
set -e
cd "$(dirname "$0")/.."
SRC=stereo_visual_odometry_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Iinclude -DHIPBA_SETUP_TIMING"
mkdir -p scripts/dev
/opt/rocm/bin/hipcc $FLAGS -c $SRC/hipba_api.hip -o scripts/dev/hipba_api_setup.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/dev/libhipba_setup.so build/lm_kernels.o build/dense_solver.o scripts/dev/hipba_api_setup.o build/rccl_comm.o build/stereo_kernels.o -ldl
echo built scripts/dev/libhipba_setup.so
