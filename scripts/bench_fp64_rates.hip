// Micro-benchmark: what fp64 rate can a gfx950 CU sustain from registers, by instruction kind?
//   v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64, v_fma_f64, and MFMA + FMA issued by the same wave.
// hipcc -O3 --offload-arch=gfx950 scripts/bench_fp64_rates.hip -o scripts/dev/bench_fp64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(double* out, int iters) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma4(double* out, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 1e-3 * i;
  double a = 1.0 + threadIdx.x * 1e-9, b = blockIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0.0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// per iteration: NM MFMAs and NF FMAs from the same wave
template <int NM, int NF>
__global__ __launch_bounds__(256) void k_mix(double* out, int iters) {
  v4d acc[NM];
  double f[NF];
  for (int i = 0; i < NM; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int i = 0; i < NF; ++i) f[i] = 1e-3 * i;
  double a = 1.0 + threadIdx.x * 1e-9, b = blockIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NF / NM; ++j) f[i * (NF / NM) + j] = __builtin_fma(f[i * (NF / NM) + j], a, b);
    }
  }
  double s = 0.0;
  for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NF; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CTRL>
__device__ __forceinline__ double rot(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// the inner loop of a 32x32 wave tile on the 4x4x4 instruction: per k-step two first operands, each also rotated by
// one, two and three lane groups (DPP), two second operands, sixteen products
template <bool DPP>
__global__ __launch_bounds__(256) void k_tile4(double* out, int iters) {
  double acc[2][2][4];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int s = 0; s < 4; ++s) acc[a][b][s] = 0.0;
  double fa[2] = {threadIdx.x * 1e-3, threadIdx.x * 2e-3}, fb[2] = {blockIdx.x * 1e-3 + 1.0, blockIdx.x * 1e-3 + 2.0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      double r[4];
      r[0] = fa[a];
      if (DPP) { r[1] = rot<0x124>(fa[a]); r[2] = rot<0x128>(fa[a]); r[3] = rot<0x12c>(fa[a]); }
      else { r[1] = r[2] = r[3] = fa[a]; }
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[a][b][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(r[s], fb[b], acc[a][b][s], 0, 0, 0);
      fa[a] += 1.0;  // (keeps the rotations inside the loop)
    }
  }
  double t = 0.0;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int s = 0; s < 4; ++s) t += acc[a][b][s];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

// sixteen products per iteration from four distinct first and four distinct second operands (loop-invariant)
__global__ __launch_bounds__(256) void k_mfma4_grid(double* out, int iters) {
  double acc[4][4];
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 1e-3 * (i + 1); b[i] = blockIdx.x * 1e-3 + i; for (int j = 0; j < 4; ++j) acc[i][j] = 0.0; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  double t = 0.0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
// the same with the operands read from LDS every iteration (no barrier: each wave reads what it wrote)
__global__ __launch_bounds__(256) void k_mfma4_lds(double* out, int iters) {
  __shared__ double sm[4][64 * 8 * 2];
  double acc[4][4];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int i = 0; i < 16; ++i) sm[w][i * 64 + l] = 1e-3 * (i + l);
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int it = 0; it < iters; ++it) {
    double a[4], b[4];
    const int o = (it & 1) * 512;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = sm[w][o + i * 64 + l]; b[i] = sm[w][o + (4 + i) * 64 + l]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  double t = 0.0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <class K>
void run(const char* name, K kern, int wgs, int iters, double flops_per_wave_iter, double* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)wgs * 4 * iters * flops_per_wave_iter;
  printf("%-28s %5d workgroups: %8.3f ms  %6.1f TFLOP/s\n", name, wgs, ms, flops / ms * 1e-9);
}

int main() {
  double* d; hipMalloc(&d, 8 * 256 * 8192);
  for (int wgs : {256, 512, 1024, 2048}) {
    run("mfma 16x16x4, 2 acc", k_mfma16<2>, wgs, 8000, 2 * 2048.0, d);
    run("mfma 16x16x4, 4 acc", k_mfma16<4>, wgs, 4000, 4 * 2048.0, d);
    run("mfma 16x16x4, 8 acc", k_mfma16<8>, wgs, 2000, 8 * 2048.0, d);
    run("mfma 4x4x4 (4 blocks), 8 acc", k_mfma4<8>, wgs, 8000, 8 * 512.0, d);
    run("mfma 4x4x4 (4 blocks), 16 acc", k_mfma4<16>, wgs, 4000, 16 * 512.0, d);
    run("v_fma_f64, 16 acc", k_fma<16>, wgs, 8000, 16 * 128.0, d);
    run("v_fma_f64, 32 acc", k_fma<32>, wgs, 4000, 32 * 128.0, d);
    run("4x4x4, 4 x 4 operand grid", k_mfma4_grid, wgs, 4000, 16 * 512.0, d);
    run("4x4x4, operands from LDS", k_mfma4_lds, wgs, 4000, 16 * 512.0, d);
    run("4x4x4 tile loop, no dpp", k_tile4<false>, wgs, 4000, 16 * 512.0, d);
    run("4x4x4 tile loop, 12 dpp", k_tile4<true>, wgs, 4000, 16 * 512.0, d);
    run("mix 4 mfma + 16 fma", k_mix<4, 16>, wgs, 4000, 4 * 2048.0 + 16 * 128.0, d);
    run("mix 4 mfma + 64 fma", k_mix<4, 64>, wgs, 2000, 4 * 2048.0 + 64 * 128.0, d);
  }
  return 0;
}
