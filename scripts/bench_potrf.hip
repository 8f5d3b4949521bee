// Developer micro-benchmark: time the panel kernel with phases switched off.
#include "../stereo_visual_odometry_amd/csrc/dense_solver.hip"
#include <cstdio>
#include <vector>
using namespace hipba;
template <int PH> float run(DevP P, double* A, int ld, double* Linv, double* LinvT, const std::vector<double>& h) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_potrf_inv<PH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t lds = (size_t)kPotrfTiles * kTile * 8 + 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float tot = 0;
  for (int it = 0; it < 20; ++it) {
    hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_potrf_inv<PH>, dim3(1), dim3(256), lds, 0, P, A, ld, 0, Linv, LinvT);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 4) tot += ms;
  }
  return tot / 16 * 1000;
}
int main() {
  const int ld = 128;
  std::vector<double> h(ld * ld, 0.0);
  for (int i = 0; i < 128; ++i) for (int j = 0; j <= i; ++j) h[j * ld + i] = (i == j) ? 200.0 + i : 1.0 / (1 + i - j);
  double *A, *Linv, *LinvT; WinState* ws; WinDesc* wd;
  hipMalloc(&A, h.size() * 8); hipMalloc(&Linv, 128 * 128 * 8); hipMalloc(&LinvT, 128 * 128 * 8);
  hipMalloc(&ws, sizeof(WinState)); hipMalloc(&wd, sizeof(WinDesc));
  WinState s{}; s.active = 1; hipMemcpy(ws, &s, sizeof(s), hipMemcpyHostToDevice);
  DevP P{}; P.ws = ws; P.wd = wd;
  printf("io only      %.1f us\n", run<0>(P, A, ld, Linv, LinvT, h));
  printf("+a           %.1f us\n", run<1>(P, A, ld, Linv, LinvT, h));
  printf("+a+b         %.1f us\n", run<3>(P, A, ld, Linv, LinvT, h));
  printf("+a+b+c       %.1f us\n", run<7>(P, A, ld, Linv, LinvT, h));
  printf("all          %.1f us\n", run<15>(P, A, ld, Linv, LinvT, h));
  // correctness of the full kernel: L L^T = A, Linv L = I
  hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_potrf_inv<15>, dim3(1), dim3(256), (size_t)kPotrfTiles * kTile * 8 + 64, 0, P, A, ld, 0, Linv, LinvT);
  std::vector<double> L(ld * ld), X(ld * ld);
  hipMemcpy(L.data(), A, L.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(X.data(), Linv, X.size() * 8, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0;
  for (int i = 0; i < 128; ++i) for (int j = 0; j <= i; ++j) {
    double s1 = 0, s2 = 0;
    for (int k = 0; k <= j; ++k) s1 += L[k * ld + i] * L[k * ld + j];
    for (int k = j; k <= i; ++k) s2 += X[k * ld + i] * L[j * ld + k];
    e1 = fmax(e1, fabs(s1 - h[j * ld + i])); e2 = fmax(e2, fabs(s2 - (i == j)));
  }
  printf("max |LL^T - A| %.3e   max |Linv L - I| %.3e\n", e1, e2);
  return 0;
}
