// dispatch_probe.hip — how long does a small high-priority kernel queue behind a chip full of big workgroups?
// (dev tool; build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o /tmp/dp && /tmp/dp)
// Emulates the dense factorisation's late phase: a "bulk" kernel with 2 workgroups per CU resident
// (VGPR allocation BV per lane, 35 KB LDS, ~50 us per workgroup) on a low-priority stream, and a chain of small
// 4-workgroup kernels (VGPR allocation SV, 17 KB LDS, ~5 us) on a high-priority stream. Question: do the small
// kernels start at once when they FIT beside the resident bulk waves (2 x BV + SV <= 512), or do they queue anyway?
#include <hip/hip_runtime.h>
#include <cstdio>

template <int TOPV>
__global__ __launch_bounds__(256) void k_spin(long long ticks) {
  extern __shared__ double sm[];
  if (TOPV == 247) asm volatile("v_mov_b32 v247, 0" ::: "v247");
  if (TOPV == 215) asm volatile("v_mov_b32 v215, 0" ::: "v215");
  if (TOPV == 95) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if (TOPV == 55) asm volatile("v_mov_b32 v55, 0" ::: "v55");
  if (TOPV == 79) asm volatile("v_mov_b32 v79, 0" ::: "v79");
  if (TOPV == 71) asm volatile("v_mov_b32 v71, 0" ::: "v71");
  if (TOPV == 63) asm volatile("v_mov_b32 v63, 0" ::: "v63");
  if (TOPV == 191) asm volatile("v_mov_b32 v191, 0" ::: "v191");
  sm[threadIdx.x] = 1.0;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}

template <int BV, int SV>
static void run(const char* label, bool with_bulk, int small_lds_kb = 17, int small_wgs = 4) {
  int lo, hi;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStream_t B, M;
  hipStreamCreateWithPriority(&B, hipStreamNonBlocking, lo);
  hipStreamCreateWithPriority(&M, hipStreamNonBlocking, hi);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int nsmall = 40;
  float best = 1e9f, worst = 0.f;
  for (int rep = 0; rep < 5; ++rep) {
    if (with_bulk) hipLaunchKernelGGL(k_spin<BV>, dim3(512 * 30), dim3(256), 35 * 1024, B, 5000LL);   // 30 rounds x 50 us
    hipLaunchKernelGGL(k_spin<SV>, dim3(1), dim3(64), 0, M, 15000LL);  // let the bulk fill the chip first (150 us)
    hipEventRecord(e0, M);
    for (int i = 0; i < nsmall; ++i) hipLaunchKernelGGL(k_spin<SV>, dim3(small_wgs), dim3(256), small_lds_kb * 1024, M, 500LL);  // 5 us each
    hipEventRecord(e1, M);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best; worst = ms > worst ? ms : worst;
  }
  printf("%-52s %2d small kernels: %.1f .. %.1f us each\n", label, nsmall, best * 1e3 / nsmall, worst * 1e3 / nsmall);
  hipStreamDestroy(B); hipStreamDestroy(M);
}

int main() {
  run<247, 95>("idle chip, small = 96 VGPR", false);
  run<247, 95>("bulk 248 VGPR (2/CU = 496), small 96", true);
  run<247, 55>("bulk 248 VGPR, small 56 (does not fit: 552)", true);
  run<215, 95>("bulk 216 VGPR (432), small 96 (does not fit: 528)", true);
  run<215, 55>("bulk 216 VGPR (432), small 56 (fits: 488)", true);
  run<215, 63>("bulk 216 VGPR (432), small 64 (496)", true);
  run<215, 71>("bulk 216 VGPR (432), small 72 (504)", true);
  run<215, 79>("bulk 216 VGPR (432), small 80 (512)", true);
  // a potrf-like single workgroup (192 VGPRs): beside ONE resident bulk workgroup it fits if its LDS is <= 125 KB
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_spin<191>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  run<247, 191>("bulk 248, one WG 192 VGPR + 120 KB LDS (needs 1 slot)", true, 120, 1);
  run<247, 191>("bulk 248, one WG 192 VGPR + 131 KB LDS (needs empty CU)", true, 131, 1);
  run<247, 191>("idle chip, one WG 192 VGPR + 131 KB LDS", false, 131, 1);
  return 0;
}
