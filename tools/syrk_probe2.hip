// syrk_probe2.hip — variants of the bulk trailing-update kernel (dev tool, GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/syrk_probe2.hip -o tools/bin/syrk_probe2 && tools/bin/syrk_probe2
// Question: the product kernel keeps the matrix pipe 66 % busy with 2 waves per SIMD that
// run the same phase pattern. Variants: NW = 4 | 8 waves per workgroup (8: wave tile 64x32, half the accumulators, up to
// 4 waves per SIMD), KC = 16 | 32, FRAG = fragments of the whole chunk loaded up front.
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int kTile = 128;

struct Args { double* M; size_t ld; int kcol0, KD, r0, nt; };

template <int NW, int KC, bool FRAG, int OCC>
__global__ __launch_bounds__(64 * NW, OCC) void k_syrk(Args g) {
  constexpr int LDT = KC + 1;
  constexpr int WGR = 2, WGC = NW / 2;
  constexpr int WTR = kTile / WGR, WTC = kTile / WGC, NMR = WTR / 16, NMC = WTC / 16;
  constexpr int NT = 64 * NW;
  const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
  const int s = (q >> 6) * 8 + xcd, inner = q & 63;
  int si = (int)((sqrt(8.0 * (double)s + 1.0) - 1.0) * 0.5);
  while ((si + 1) * (si + 2) / 2 <= s) ++si;
  while (si * (si + 1) / 2 > s) --si;
  const int sj = s - si * (si + 1) / 2;
  const int ti = si * 8 + (inner >> 3), tj = sj * 8 + (inner & 7);
  if (ti >= g.nt || tj > ti) return;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double (*sA)[LDT] = reinterpret_cast<double (*)[LDT]>(smem);
  double (*sB)[LDT] = reinterpret_cast<double (*)[LDT]>(smem + kTile * LDT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave / WGC, wc = wave % WGC;
  const size_t ld = g.ld;
  const char* Ag = reinterpret_cast<const char*>(g.M + (size_t)(g.r0 + ti * kTile) * ld + g.kcol0);
  const char* Bg = reinterpret_cast<const char*>(g.M + (size_t)(g.r0 + tj * kTile) * ld + g.kcol0);
  char* Cg = reinterpret_cast<char*>(g.M + (size_t)(g.r0 + ti * kTile) * ld + (size_t)(g.r0 + tj * kTile));
  const unsigned ldab = (unsigned)(ld * sizeof(double));
  constexpr int LPR = KC / 2, RPP = NT / LPR, NPA = kTile / RPP;
  const int c2 = (tid % LPR) * 2, rbase = tid / LPR;
  const unsigned offa = (unsigned)rbase * ldab + (unsigned)c2 * 8u;
  double2 pa[NPA], pb[NPA];
  auto gload = [&](int kc) {
    const char* Ak = Ag + (size_t)kc * 8; const char* Bk = Bg + (size_t)kc * 8;
#pragma unroll
    for (int it = 0; it < NPA; ++it) { pa[it] = *reinterpret_cast<const double2*>(Ak + (size_t)(RPP * it) * ldab + offa); pb[it] = *reinterpret_cast<const double2*>(Bk + (size_t)(RPP * it) * ldab + offa); }
  };
  gload(0);
  const int fr = lane & 15, fk = lane >> 4;
  const unsigned offc = (unsigned)(wr * WTR + fk) * ldab + (unsigned)(wc * WTC + fr) * 8u;
  v4f64 acc[NMR][NMC];
#pragma unroll
  for (int tm = 0; tm < NMR; ++tm)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int tn = 0; tn < NMC; ++tn) acc[tm][tn][rg] = *reinterpret_cast<const double*>(Cg + offc + (unsigned)(tm * 16 + 4 * rg) * ldab + tn * 128);
  for (int kc = 0; kc < g.KD; kc += KC) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NPA; ++it) {
      sA[rbase + RPP * it][c2] = -pa[it].x; sA[rbase + RPP * it][c2 + 1] = -pa[it].y;
      sB[rbase + RPP * it][c2] = pb[it].x; sB[rbase + RPP * it][c2 + 1] = pb[it].y;
    }
    __syncthreads();
    if (kc + KC < g.KD) gload(kc + KC);
    if (FRAG) {
      double a[KC / 4][NMR], bb[KC / 4][NMC];
#pragma unroll
      for (int k4 = 0; k4 < KC / 4; ++k4) {
#pragma unroll
        for (int t = 0; t < NMR; ++t) a[k4][t] = sA[wr * WTR + t * 16 + fr][4 * k4 + fk];
#pragma unroll
        for (int t = 0; t < NMC; ++t) bb[k4][t] = sB[wc * WTC + t * 16 + fr][4 * k4 + fk];
      }
#pragma unroll
      for (int k4 = 0; k4 < KC / 4; ++k4)
#pragma unroll
        for (int tm = 0; tm < NMR; ++tm)
#pragma unroll
          for (int tn = 0; tn < NMC; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k4][tm], bb[k4][tn], acc[tm][tn], 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < KC; kk += 4) {
        double a[NMR], bb[NMC];
#pragma unroll
        for (int t = 0; t < NMR; ++t) a[t] = sA[wr * WTR + t * 16 + fr][kk + fk];
#pragma unroll
        for (int t = 0; t < NMC; ++t) bb[t] = sB[wc * WTC + t * 16 + fr][kk + fk];
#pragma unroll
        for (int tm = 0; tm < NMR; ++tm)
#pragma unroll
          for (int tn = 0; tn < NMC; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], bb[tn], acc[tm][tn], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int tm = 0; tm < NMR; ++tm)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int tn = 0; tn < NMC; ++tn) *reinterpret_cast<double*>(Cg + offc + (unsigned)(tm * 16 + 4 * rg) * ldab + tn * 128) = acc[tm][tn][rg];
}

template <int NW, int KC, bool FRAG, int OCC>
void run(const char* name, double* M, size_t ld, hipEvent_t e0, hipEvent_t e1) {
  const size_t lds = (size_t)2 * kTile * (KC + 1) * sizeof(double);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk<NW, KC, FRAG, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int nt : {25, 50, 100}) {
    const int KD = 256, tb = 4;
    const int Ts = (nt + 7) / 8, ns = Ts * (Ts + 1) / 2, nblk = ((ns + 7) / 8) * 8 * 64;
    Args g{M, ld, 0, KD, tb * kTile, nt};
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k_syrk<NW, KC, FRAG, OCC>), dim3(nblk), dim3(64 * NW), lds, 0, g);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    const double flops = (double)nt * (nt + 1) / 2 * 2.0 * kTile * kTile * KD;
    printf("%-34s nt %3d : %7.1f us  %5.1f TFLOP/s\n", name, nt, best * 1e3, flops / (best * 1e-3) / 1e12);
  }
}

int main() {
  const int ntmax = 100, kpan = 512;
  const size_t ld = (size_t)(ntmax * kTile + kpan);
  double* M; hipMalloc(&M, ld * ld * sizeof(double)); hipMemset(M, 0, ld * ld * sizeof(double));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  run<4, 16, false, 2>("4 waves KC16 (product shape)", M, ld, e0, e1);
  run<4, 16, true, 2>("4 waves KC16 frags up front", M, ld, e0, e1);
  run<4, 32, true, 2>("4 waves KC32 frags up front", M, ld, e0, e1);
  run<8, 16, false, 1>("8 waves KC16 occ1", M, ld, e0, e1);
  run<8, 16, false, 2>("8 waves KC16 occ2", M, ld, e0, e1);
  run<8, 16, true, 2>("8 waves KC16 frags occ2", M, ld, e0, e1);
  run<8, 32, false, 2>("8 waves KC32 occ2", M, ld, e0, e1);
  run<8, 32, true, 2>("8 waves KC32 frags occ2", M, ld, e0, e1);
  return 0;
}
