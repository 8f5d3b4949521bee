// Probe of the lane layout of v_mfma_f64_4x4x4_4b_f64: which (A lane, B lane) pairs feed which D lane.
// hipcc -O3 --offload-arch=gfx950 scripts/probe_mfma4x4.hip -o scripts/dev/probe_mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double* out) {
  const int lane = threadIdx.x;
  for (int p = 0; p < 64; ++p)
    for (int q = 0; q < 64; ++q) {
      const double a = lane == p ? 1.0 : 0.0, b = lane == q ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      out[(p * 64 + q) * 64 + lane] = d;
    }
}
int main() {
  double* d; hipMalloc(&d, 8 * 64 * 64 * 64);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  std::vector<double> h(64 * 64 * 64);
  hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
  // for every D lane: the list of (p, q) that feed it
  for (int L = 0; L < 64; ++L) {
    printf("D lane %2d <-", L);
    for (int p = 0; p < 64; ++p)
      for (int q = 0; q < 64; ++q)
        if (h[(p * 64 + q) * 64 + L] != 0.0) printf(" (a%d,b%d)", p, q);
    printf("\n");
  }
  return 0;
}
