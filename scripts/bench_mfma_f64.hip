// Micro-benchmark: attainable v_mfma_f64_16x16x4_f64 rate on the whole chip (register operands only).
// hipcc -O3 --offload-arch=gfx950 scripts/bench_mfma_f64.hip -o scripts/dev/bench_mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[8 * 256 * 4096 / 8 - 2] = (double)(c1 - c0);
    out[8 * 256 * 4096 / 8 - 1] = (double)(w1 - w0);
  }
}
template <int NACC>
void run(int wgs, int iters, double* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(256), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)wgs * 4 * iters * NACC * 2048.0;
  double cw[2];
  hipMemcpy(cw, d + 8 * 256 * 4096 / 8 - 2, 16, hipMemcpyDeviceToHost);
  printf("acc tiles %2d, %5d workgroups: %.3f ms, %.1f TFLOP/s; clock64 %.0f ticks per 100 MHz tick %.0f -> %.0f MHz; %.1f cycles per MFMA per SIMD\n",
         NACC, wgs, ms, flops / ms * 1e-9, cw[0], cw[1], cw[0] / cw[1] * 100.0, cw[0] / ((double)iters * NACC * (wgs > 256 ? (wgs / 256) : 1)));
}
int main() {
  double* d; hipMalloc(&d, 8 * 256 * 4096);
  run<16>(256, 2000, d);
  run<16>(512, 2000, d);
  run<16>(1024, 2000, d);
  run<4>(512, 8000, d);
  run<4>(1024, 8000, d);
  run<1>(1024, 16000, d);
  run<1>(2048, 16000, d);
  return 0;
}
