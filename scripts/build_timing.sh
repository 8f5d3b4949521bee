#!/bin/bash
# Development build with workgroup phase timestamps (HIPBA_PHASE_TIMING) -> scripts/dev/libhipba_timing.so
set -e
cd "$(dirname "$0")/.."
SRC=stereo_visual_odometry_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Iinclude -DHIPBA_PHASE_TIMING -fgpu-rdc"
mkdir -p scripts/dev
/opt/rocm/bin/hipcc $FLAGS -shared -o scripts/dev/libhipba_timing.so $SRC/lm_kernels.hip $SRC/dense_solver.hip $SRC/hipba_api.hip $SRC/rccl_comm.hip $SRC/stereo_kernels.hip -ldl
echo built scripts/dev/libhipba_timing.so
