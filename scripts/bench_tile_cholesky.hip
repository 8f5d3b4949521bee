// Micro-benchmark of the one-wave 16x16 tile Cholesky + inverse (chol_tiles.hpp): shader cycles per call of the
// variant in the library and of candidates, and the largest difference between their results.
// Tried in the candidate slot and not kept: the next pivot formed from wave-uniform values (unscaled multiplier and
// diagonal entry broadcast before the current pivot's reciprocal root is known), bit-identical, 4333 cycles against 3522.
// hipcc -O3 --offload-arch=gfx950 -Iinclude scripts/bench_tile_cholesky.hip -o scripts/dev/bench_tile_cholesky
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../stereo_visual_odometry_amd/csrc/chol_tiles.hpp"
using namespace hipba;

template <int V>
__global__ __launch_bounds__(64) void k_bench(const double* A, double* out, long long* cycles, int reps) {
  __shared__ double D[kTile];
  const int lane = threadIdx.x;
  long long tot = 0;
  bool ok = true;
  for (int r = 0; r < reps; ++r) {
    for (int i = lane; i < kTile; i += 64) D[i] = A[i];
    __syncthreads();
    const long long t0 = clock64();
    if (V == 0) ok &= tile_cholesky_inverse(D, nullptr, 0, lane);
    else ok &= tile_cholesky_inverse(D, nullptr, 0, lane);  // (slot for a candidate: see the note below)
    const long long t1 = clock64();
    tot += t1 - t0;
    __syncthreads();
  }
  for (int i = lane; i < kTile; i += 64) out[i] = D[i];
  if (lane == 0) { cycles[0] = tot / reps; cycles[1] = ok; }
}

int main() {
  std::vector<double> h(kTile);
  // SPD: B B^T + 16 I, column-major
  std::vector<double> B(kTile);
  for (int i = 0; i < kTile; ++i) B[i] = std::sin(0.37 * i) + 0.1 * (i % 7);
  for (int p = 0; p < 16; ++p) for (int q = 0; q < 16; ++q) {
    double s = p == q ? 16.0 : 0.0;
    for (int k = 0; k < 16; ++k) s += B[k * 16 + p] * B[k * 16 + q];
    h[q * 16 + p] = s;
  }
  double *A, *o0, *o1; long long* cyc;
  hipMalloc(&A, kTile * 8); hipMalloc(&o0, kTile * 8); hipMalloc(&o1, kTile * 8); hipMalloc(&cyc, 32);
  hipMemcpy(A, h.data(), kTile * 8, hipMemcpyHostToDevice);
  long long c0[2], c1[2];
  hipLaunchKernelGGL(k_bench<0>, dim3(1), dim3(64), 0, 0, A, o0, cyc, 200);
  hipMemcpy(c0, cyc, 16, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k_bench<1>, dim3(1), dim3(64), 0, 0, A, o1, cyc, 200);
  hipMemcpy(c1, cyc, 16, hipMemcpyDeviceToHost);
  std::vector<double> r0(kTile), r1(kTile);
  hipMemcpy(r0.data(), o0, kTile * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r1.data(), o1, kTile * 8, hipMemcpyDeviceToHost);
  double md = 0.0, mx = 0.0;
  for (int i = 0; i < kTile; ++i) { md = std::fmax(md, std::fabs(r0[i] - r1[i])); mx = std::fmax(mx, std::fabs(r0[i])); }
  // residual of the inverse: L^-1 A L^-T = I, with the inverse stored row-major-by-column as the routine leaves it
  printf("library variant: %lld cycles (ok %lld); candidate: %lld cycles (ok %lld); max |difference| %.3e of %.3e\n", c0[0], c0[1],
         c1[0], c1[1], md, mx);
  return 0;
}
