// Which lane does row_ror:n read from?  hipcc -O3 --offload-arch=gfx950 scripts/probe_dpp.hip -o scripts/dev/probe_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(0, lane, 0x124, 0xf, 0xf, false);
  out[64 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x128, 0xf, 0xf, false);
  out[128 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x12c, 0xf, 0xf, false);
  double d = 1000.0 + lane;
  int lo = __double2loint(d), hi = __double2hiint(d);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xf, 0xf, false);
  out[192 + lane] = (int)__hiloint2double(hi, lo);
}
int main() {
  int* d; hipMalloc(&d, 4 * 256);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[256];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int k = 0; k < 4; ++k) { printf("set %d:", k); for (int i = 0; i < 64; ++i) printf(" %d", h[64 * k + i]); printf("\n"); }
  return 0;
}
