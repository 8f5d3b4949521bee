// Does v_mfma_f64_4x4x4_4b_f64 honour the A-block broadcast controls (cbsz / abid)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CBSZ, int ABID>
__global__ void probe(double* out) {
  const int lane = threadIdx.x;
  for (int p = 0; p < 64; ++p)
    for (int q = 0; q < 64; ++q) {
      const double a = lane == p ? 1.0 : 0.0, b = lane == q ? 1.0 : 0.0;
      out[(p * 64 + q) * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
    }
}
template <int CBSZ, int ABID>
void show(double* d) {
  hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d);
  std::vector<double> h(64 * 64 * 64);
  hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
  printf("cbsz %d abid %d\n", CBSZ, ABID);
  for (int L : {0, 1, 5, 16, 21, 37, 63}) {
    printf("  D lane %2d <-", L);
    for (int p = 0; p < 64; ++p)
      for (int q = 0; q < 64; ++q)
        if (h[(p * 64 + q) * 64 + L] != 0.0) printf(" (a%d,b%d)", p, q);
    printf("\n");
  }
}
int main() {
  double* d; hipMalloc(&d, 8 * 64 * 64 * 64);
  show<0, 0>(d);
  show<2, 0>(d);
  show<2, 1>(d);
  show<2, 3>(d);
  show<1, 1>(d);
  return 0;
}
