// diag_probe.hip — the 16x16 diagonal-block factorisation (+ inverse) of k_potrf_panel in isolation, variants timed with
// s_memtime on one wave (dev tool): hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/diag_probe.hip -o /tmp/diag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
constexpr int PB = 16, PP = 17;
__device__ __forceinline__ long long tick(double& x) {
  long long t; int dummy;
  int lo = (int)__double_as_longlong(x);
  asm volatile("v_readfirstlane_b32 %1, %2\n\ts_add_u32 %1, %1, 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(dummy) : "v"(lo) : "memory", "scc");
  asm volatile("" : "+v"(x));
  return t;
}
// VAR 0: as in k_panel.hip v2 | 1: no inverse part | 2: no rsq part (scale factors from saved pivots at the end) | 3: 1+2
template <int VAR>
__device__ __forceinline__ void diag_block(double* cur, double* colA, double* rowW, double* sDv, double* sdd, int lane) {
  const int r = lane >> 2, q = lane & 3;
  double a[4], w[4], rsc[4], rsr = 0.0;
#pragma unroll
  for (int e = 0; e < 4; ++e) { a[e] = cur[r * PP + 4 * q + e]; w[e] = (4 * q + e == r) ? 1.0 : 0.0; rsc[e] = 0.0; }
#pragma unroll
  for (int c = 0; c < PB; ++c) {
    const int qc = c >> 2, ec = c & 3;
    if (q == qc) colA[r] = a[ec];
    if (!(VAR & 1)) if (r == c) {
#pragma unroll
      for (int e = 0; e < 4; ++e) rowW[4 * q + e] = w[e];
    }
    __builtin_amdgcn_wave_barrier();
    const double d = colA[c];
    const double mr = colA[r];
    double cv[4], xr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { cv[e] = colA[4 * q + e]; if (!(VAR & 1)) xr[e] = rowW[4 * q + e]; }
    __builtin_amdgcn_wave_barrier();
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e0 = fma(-d, r0, 1.0);
    const double r1 = fma(r0, e0, r0), e1 = e0 * e0;
    const double rinv = fma(r1, e1, r1);
    const double t = mr * rinv;
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] -= ((4 * q + e > c) ? t : 0.0) * cv[e];
    if (!(VAR & 1)) {
      const double tw = (r > c) ? t : 0.0;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] -= tw * xr[e];
    }
    if (!(VAR & 2)) {
      double rs = __builtin_amdgcn_rsq(d);
      rs = rs * (1.5 - 0.5 * d * rs * rs);
      rs = rs * (1.5 - 0.5 * d * rs * rs);
      if (q == qc) rsc[ec] = rs;
      if (r == c) rsr = rs;
    } else if (lane == 0) sdd[c] = d;
  }
  if (VAR & 2) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < 4; ++e) { const double d = sdd[4 * q + e]; double rs = __builtin_amdgcn_rsq(d); rs = rs * (1.5 - 0.5 * d * rs * rs); rsc[e] = rs * (1.5 - 0.5 * d * rs * rs); }
    { const double d = sdd[r]; double rs = __builtin_amdgcn_rsq(d); rs = rs * (1.5 - 0.5 * d * rs * rs); rsr = rs * (1.5 - 0.5 * d * rs * rs); }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool low = 4 * q + e <= r;
    cur[r * PP + 4 * q + e] = low ? a[e] * rsc[e] : 0.0;
    if (!(VAR & 1)) sDv[r * PB + 4 * q + e] = low ? w[e] * rsr : 0.0;
  }
}
// VAR 5: L only in the sweep (scale factors at the end), then X = L^-1 by column-parallel forward substitution: lane c owns
// column c, every L entry is a broadcast LDS read, no cross-lane traffic
__device__ __forceinline__ void inverse_after(const double* cur, const double* sinv, double* sDv, int lane) {
  const int c = lane & 15;
  double x[PB];
#pragma unroll
  for (int r = 0; r < PB; ++r) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < r; k += 2) { s0 += cur[r * PP + k] * x[k]; if (k + 1 < r) s1 += cur[r * PP + k + 1] * x[k + 1]; }
    const double iv = sinv[r];
    x[r] = (r == c) ? iv : (r > c ? -(s0 + s1) * iv : 0.0);
  }
  if (lane < PB) {
#pragma unroll
    for (int r = 0; r < PB; ++r) sDv[r * PB + c] = x[r];
  }
}
// VAR 4: the round-2a formulation (row per lane, v_readlane broadcasts), no inverse
__device__ __forceinline__ double rdl(double v, int l) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ void diag_block_rl(double* cur, int lane) {
  const int r = lane & 15;
  double x[PB];
#pragma unroll
  for (int c = 0; c < PB; ++c) x[c] = (c <= r) ? cur[r * PP + c] : 0.0;
#pragma unroll
  for (int c = 0; c < PB; ++c) {
    const double d = rdl(x[c], c);
    double inv = __builtin_amdgcn_rsq(d);
    inv = inv * (1.5 - 0.5 * d * inv * inv);
    inv = inv * (1.5 - 0.5 * d * inv * inv);
    x[c] = (r == c) ? d * inv : x[c] * inv;
#pragma unroll
    for (int cc = c + 1; cc < PB; ++cc) x[cc] -= x[c] * rdl(x[c], cc);
  }
  if (lane < PB) {
#pragma unroll
    for (int c = 0; c < PB; ++c) cur[r * PP + c] = (c <= r) ? x[c] : 0.0;
  }
}
template <int VAR>
__global__ __launch_bounds__(64) void k_diag(const double* A, double* Lout, double* Xout, long long* cyc) {
  __shared__ double cur[16 * PP], colA[16], rowW[16], sDv[256], sdd[16], keep[16 * PP];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) keep[(i >> 4) * PP + (i & 15)] = A[i];
  __syncthreads();
  long long best = 1ll << 60;
  for (int rep = 0; rep < 20; ++rep) {
    for (int i = lane; i < 16 * PP; i += 64) cur[i] = keep[i];
    __syncthreads();
    double x = cur[lane];
    const long long t0 = tick(x);
    if (VAR == 4) diag_block_rl(cur, lane);
    else if (VAR == 5) {
      diag_block<3>(cur, colA, rowW, sDv, sdd, lane);
      __builtin_amdgcn_wave_barrier();
      if (lane < 16) { double d = sdd[lane]; double rs = __builtin_amdgcn_rsq(d); rs = rs * (1.5 - 0.5 * d * rs * rs); colA[lane] = rs * (1.5 - 0.5 * d * rs * rs); }
      __builtin_amdgcn_wave_barrier();
      inverse_after(cur, colA, sDv, lane);
    } else diag_block<VAR>(cur, colA, rowW, sDv, sdd, lane);
    __syncthreads();
    x = cur[lane] + x;
    const long long t1 = tick(x);
    if (t1 - t0 < best) best = t1 - t0;
    if (x == 1234.5) cur[0] = x;
  }
  for (int i = lane; i < 256; i += 64) { Lout[i] = cur[(i >> 4) * PP + (i & 15)]; Xout[i] = sDv[i]; }
  if (lane == 0) cyc[0] = best;
}
int main() {
  std::vector<double> A(256), B(256);
  srand(3);
  for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { double s = (r == c) ? 0.5 : 0.0; for (int k = 0; k < 16; ++k) s += B[r * 16 + k] * B[c * 16 + k]; A[r * 16 + c] = s; }
  std::vector<double> H(A);
  for (int j = 0; j < 16; ++j) { double d = H[j * 16 + j]; for (int k = 0; k < j; ++k) d -= H[j * 16 + k] * H[j * 16 + k]; d = sqrt(d); H[j * 16 + j] = d;
    for (int i = j + 1; i < 16; ++i) { double s = H[i * 16 + j]; for (int k = 0; k < j; ++k) s -= H[i * 16 + k] * H[j * 16 + k]; H[i * 16 + j] = s / d; } }
  double *dA, *dL, *dX; long long* dc;
  hipMalloc(&dA, 2048); hipMalloc(&dL, 2048); hipMalloc(&dX, 2048); hipMalloc(&dc, 64);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
  const char* nm[6] = {"LDS-column form, L + inverse (k_panel v2)", "LDS-column form, L only", "v2 with scale factors at the end", "L only, scale factors at the end", "row per lane + v_readlane, L only (round 2a)", "L only in the sweep, inverse by column substitution after it"};
  for (int v = 0; v < 6; ++v) {
    hipMemset(dX, 0, 2048);
    switch (v) {
      case 0: hipLaunchKernelGGL(k_diag<0>, dim3(1), dim3(64), 0, 0, dA, dL, dX, dc); break;
      case 1: hipLaunchKernelGGL(k_diag<1>, dim3(1), dim3(64), 0, 0, dA, dL, dX, dc); break;
      case 2: hipLaunchKernelGGL(k_diag<2>, dim3(1), dim3(64), 0, 0, dA, dL, dX, dc); break;
      case 3: hipLaunchKernelGGL(k_diag<3>, dim3(1), dim3(64), 0, 0, dA, dL, dX, dc); break;
      case 4: hipLaunchKernelGGL(k_diag<4>, dim3(1), dim3(64), 0, 0, dA, dL, dX, dc); break;
      case 5: hipLaunchKernelGGL(k_diag<5>, dim3(1), dim3(64), 0, 0, dA, dL, dX, dc); break;
    }
    hipDeviceSynchronize();
    std::vector<double> L(256), X(256); long long c;
    hipMemcpy(L.data(), dL, 2048, hipMemcpyDeviceToHost); hipMemcpy(X.data(), dX, 2048, hipMemcpyDeviceToHost); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    double eL = 0, eX = 0;
    for (int r = 0; r < 16; ++r) for (int cc = 0; cc <= r; ++cc) eL = fmax(eL, fabs(L[r * 16 + cc] - H[r * 16 + cc]));
    if ((!(v & 1) && v != 4) || v == 5) for (int r = 0; r < 16; ++r) for (int cc = 0; cc < 16; ++cc) { double s = 0; for (int k = 0; k < 16; ++k) s += X[r * 16 + k] * (k >= cc ? H[k * 16 + cc] : 0.0); eX = fmax(eX, fabs(s - (r == cc))); }
    printf("%-48s %6lld cycles = %.2f us   max|L-L_host| %.2e  max|X L - I| %.2e\n", nm[v], c, c / 2400.0, eL, eX);
  }
  return 0;
}
