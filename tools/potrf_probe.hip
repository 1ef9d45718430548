// potrf_probe.hip — phase timing of k_potrf_inv on an idle GPU (dev tool; build + run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCOVGPU_PROBE tools/potrf_probe.hip -o /tmp/potrf_probe && /tmp/potrf_probe
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../covins_amd/csrc/k_chol.hip"
#include "../covins_amd/csrc/k_panel.hip"  // (k_chol.hip schedules its kernels)

using namespace covgpu;

int main() {
  const int n = 128;
  std::vector<double> A(n * n, 0.0);
  for (int r = 0; r < n; ++r) for (int c = 0; c <= r; ++c) A[r * n + c] = (r == c) ? 300.0 + r : 1.0 / (1 + r - c);
  double *dA, *dL; int* df;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&df, 16);
  hipMemset(df, 0, 16);
  const size_t lds = (size_t)kTile * (kTile + 1) * sizeof(double);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_inv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  const int NIT = 200;
  std::vector<std::vector<double>> ph(5);
  std::vector<double> tot;
  for (int it = 0; it < NIT; ++it) {
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    { long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof(z)); }
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_potrf_inv, dim3(1), dim3(256), lds, 0, dA, (size_t)n, 0, dL, df, (const double*)nullptr, (double*)nullptr, (size_t)0, (size_t)0, (size_t)0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    long long pr[8];
    hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe), sizeof(pr));
    for (int k = 0; k < 5; ++k) ph[k].push_back((pr[k + 1] - pr[k]) / 100.0);
    tot.push_back((pr[5] - pr[0]) / 100.0);
  }
  auto stat = [](std::vector<double> v, double& mn, double& med) { std::sort(v.begin(), v.end()); mn = v.front(); med = v[v.size() / 2]; };
  const char* names[5] = {"load", "chol", "toLDS/writeback", "inverse", "store"};
  printf("potrf_inv: best event time %.1f us over %d runs (100 MHz wall clock inside the kernel):\n", best * 1e3, NIT);
  for (int k = 0; k < 5; ++k) { double mn, med; stat(ph[k], mn, med); printf("  %-16s min %.1f  median %.1f us\n", names[k], mn, med); }
  { double mn, med; stat(tot, mn, med); printf("  %-16s min %.1f  median %.1f us\n", "kernel body", mn, med); }
  {  // the blocked kernel (COVGPU_POTRF=2): same matrix, phases load | (a)(b)(c) loop | L store + diag copy | inverse levels | store
    const size_t lds2 = lds + (size_t)(8 * 16 * 16 + kTile) * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_inv_blk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    std::vector<std::vector<double>> q(8);
    float best2 = 1e9;
    for (int it = 0; it < NIT; ++it) {
      hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
      { long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof(z)); }
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_potrf_inv_blk, dim3(1), dim3(256), lds2, 0, dA, (size_t)n, 0, dL, df, (const double*)nullptr, (double*)nullptr, (size_t)0, (size_t)0, (size_t)0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best2) best2 = ms;
      long long pr[8];
      hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe), sizeof(pr));
      q[0].push_back((pr[1] - pr[0]) / 100.0); q[1].push_back((pr[2] - pr[1]) / 100.0); q[2].push_back((pr[3] - pr[2]) / 100.0);
      q[3].push_back((pr[4] - pr[3]) / 100.0); q[4].push_back(0.0);
      q[5].push_back(pr[5] / 100.0); q[6].push_back(pr[6] / 100.0); q[7].push_back(pr[7] / 100.0);
    }
    const char* nm[8] = {"load", "chol loop", "L store+diag", "inverse levels", "-", "  (a) diag blocks", "  (b) panel rows", "  (c) trailing MFMA"};
    printf("potrf_inv_blk: best event time %.1f us\n", best2 * 1e3);
    for (int k = 0; k < 8; ++k) { if (k == 4) continue; double mn, med; stat(q[k], mn, med); printf("  %-20s min %.1f  median %.1f us\n", nm[k], mn, med); }
  }
  // correctness on the last run: || L L^T - A ||_max and || Linv L - I ||_max
  std::vector<double> L(n * n), Li(n * n);
  hipMemcpy(L.data(), dA, n * n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(Li.data(), dL, n * n * 8, hipMemcpyDeviceToHost);
  double e_llt = 0, e_inv = 0;
  for (int r = 0; r < n; ++r)
    for (int c = 0; c <= r; ++c) {
      double s1 = 0, s2 = 0;
      for (int k = 0; k <= c; ++k) s1 += L[r * n + k] * L[c * n + k];
      for (int k = c; k <= r; ++k) s2 += Li[r * n + k] * L[k * n + c];
      e_llt = fmax(e_llt, fabs(s1 - A[r * n + c]));
      e_inv = fmax(e_inv, fabs(s2 - (r == c ? 1.0 : 0.0)));
    }
  int fl = 0; hipMemcpy(&fl, df, 4, hipMemcpyDeviceToHost);
  printf("  max|LL^T-A| %.3e  max|Linv L - I| %.3e  flag %d\n", e_llt, e_inv, fl);
  return 0;
}
