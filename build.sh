#!/bin/bash
# Builds libhipba.so (gfx950 only) in-tree; hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
SRC=stereo_visual_odometry_amd/csrc
OUT=stereo_visual_odometry_amd/lib
mkdir -p "$OUT" build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Iinclude"
pids=()
for f in lm_kernels dense_solver hipba_api rccl_comm stereo_kernels; do
  if [ ! -f build/$f.o ] || [ $SRC/$f.hip -nt build/$f.o ] || [ -n "$(find $SRC include -name '*.h*' -newer build/$f.o | head -1)" ]; then
    $HIPCC $FLAGS -c $SRC/$f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libhipba.so build/lm_kernels.o build/dense_solver.o build/hipba_api.o build/rccl_comm.o build/stereo_kernels.o -ldl
echo "built $OUT/libhipba.so"
# C++ host adapter demo: the ROS-free ba_test harness (plain g++, links the C ABI only)
g++ -O2 -std=c++17 -Wall -Iinclude -I$SRC/host -o $OUT/ba_test $SRC/host/ba_test_cli.cpp -L$OUT -lhipba -Wl,-rpath,'$ORIGIN'
echo "built $OUT/ba_test"
g++ -O2 -std=c++17 -Wall -Iinclude -I$SRC/host -o $OUT/ba_replay $SRC/host/ba_replay_cli.cpp -L$OUT -lhipba -Wl,-rpath,'$ORIGIN'
echo "built $OUT/ba_replay"
g++ -O2 -std=c++17 -Wall -pthread -Iinclude -I$SRC/host -o $OUT/ba_stage_demo $SRC/host/ba_stage_demo.cpp -L$OUT -lhipba -Wl,-rpath,'$ORIGIN'
echo "built $OUT/ba_stage_demo"
g++ -O2 -std=c++17 -Wall -Iinclude -I$SRC/host -o $OUT/kitti_info $SRC/host/kitti_info_cli.cpp
echo "built $OUT/kitti_info"
g++ -O2 -std=c++17 -Wall -pthread -Iinclude -I$SRC/host -o $OUT/ba_sequences $SRC/host/ba_sequences_cli.cpp -L$OUT -lhipba -Wl,-rpath,'$ORIGIN'
echo "built $OUT/ba_sequences"
g++ -O2 -std=c++17 -Wall -pthread -Iinclude -I$SRC/host -o $OUT/host_selftest $SRC/host/host_selftest.cpp -L$OUT -lhipba -Wl,-rpath,'$ORIGIN'
echo "built $OUT/host_selftest"
g++ -O2 -std=c++17 -Wall -pthread -Iinclude -I$SRC/host -o $OUT/kitti_runner $SRC/host/kitti_runner_cli.cpp -L$OUT -lhipba -Wl,-rpath,'$ORIGIN'
echo "built $OUT/kitti_runner"
g++ -O2 -std=c++17 -Wall -Iinclude -I$SRC/host -o $OUT/frontend_steps $SRC/host/frontend_steps_cli.cpp -L$OUT -lhipba -Wl,-rpath,'$ORIGIN'
echo "built $OUT/frontend_steps"
g++ -O2 -std=c++17 -Wall -Iinclude -I$SRC/host -o $OUT/vo_pipeline $SRC/host/vo_pipeline_cli.cpp -L$OUT -lhipba -Wl,-rpath,'$ORIGIN'
echo "built $OUT/vo_pipeline"
