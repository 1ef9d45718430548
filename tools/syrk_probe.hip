// syrk_probe.hip — throughput of the bulk trailing-update kernel alone, as a function of the update rank K and of the
// triangle order (dev tool; next-round question: would rank-512 panels pay for the large early updates?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/syrk_probe.hip -o /tmp/syrk_probe && /tmp/syrk_probe
#include <cstdio>
#include <vector>
#include "../covins_amd/csrc/k_chol.hip"
#include "../covins_amd/csrc/k_panel.hip"  // (k_chol.hip schedules its kernels)

using namespace covgpu;

int main() {
  const int ntmax = 100, kpan = 512;
  const size_t ld = (size_t)(ntmax * kTile + kpan);
  double* M;
  hipMalloc(&M, ld * ld * sizeof(double));
  hipMemset(M, 0, ld * ld * sizeof(double));
  const size_t lds_gemm = (size_t)2 * kTile * LDT * sizeof(double);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_abt<MODE_SYRK_TRI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_gemm);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nt : {25, 50, 75, 100}) {
    for (int KD : {128, 256, 512}) {
      const int tb = kpan / kTile;  // triangle starts right of the panel columns
      const int Ts = (nt + 7) / 8, ns = Ts * (Ts + 1) / 2, nblk = ((ns + 7) / 8) * 8 * 64;
      GemmArgs g{M, ld, 0, KD, tb * kTile, tb * kTile, tb * kTile, nt, nullptr, nullptr, nullptr, 0, 0, 0};
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gemm_abt<MODE_SYRK_TRI>, dim3(nblk), dim3(256), lds_gemm, 0, g);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double flops = (double)nt * (nt + 1) / 2 * 2.0 * kTile * kTile * KD;
      printf("nt %3d  rank %3d : %7.1f us  %5.1f TFLOP/s\n", nt, KD, best * 1e3, flops / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
