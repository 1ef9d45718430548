// potrf_probe.hip — phase timing of k_potrf_inv on an idle GPU (dev tool; build + run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCOVGPU_PROBE tools/potrf_probe.hip -o /tmp/potrf_probe && /tmp/potrf_probe
#include <cstdio>
#include <vector>
#include "../covins_amd/csrc/k_chol.hip"

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
  for (int it = 0; it < 20; ++it) {
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    { long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof(z)); }
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_potrf_inv, dim3(1), dim3(256), lds, 0, dA, (size_t)n, 0, dL, df);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  long long pr[8];
  hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe), sizeof(pr));
  printf("potrf_inv best %.1f us; phases (100 MHz ticks -> us): load %.1f chol %.1f toLDS %.1f inverse %.1f store %.1f\n", best * 1e3,
         (pr[1] - pr[0]) / 100.0, (pr[2] - pr[1]) / 100.0, (pr[3] - pr[2]) / 100.0, (pr[4] - pr[3]) / 100.0, (pr[5] - pr[4]) / 100.0);
  printf("  inside chol: U %.1f us  D %.1f us  (T = rest)\n", pr[6] / 100.0, pr[7] / 100.0);
  return 0;
}
