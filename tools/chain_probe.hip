// chain_probe.hip — the serial speed-bias chain kernels (k_struct.hip) alone on an idle GPU (dev tool):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCOVGPU_PROBE tools/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
// One fake IMU chain of NPOS positions with well-conditioned 9x9 blocks; checks the block-bidiagonal factor against a host
// recurrence and times k_sb_chain_factor / k_sb_backsolve with HIP events (1 and 5 chains).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../covins_amd/csrc/k_struct.hip"
namespace covgpu {  // symbols k_struct.hip's launchers reference (never called here)
void dense_cholesky_solve_raw(double*, double*, double*, int*, int, hipStream_t, CholAux&, int, bool, DenseBatch) {}
void launch_arrow_solve(const DevProblem&, hipStream_t, CholAux&) {}
void launch_pgo_block_solve(const DevProblem&, PgoPlan&, hipStream_t, CholAux&) {}
void CholAux::init() {}
}
using namespace covgpu;

int main() {
  const int NCH = 5, NPOS = 500, K = NCH * NPOS;
  std::mt19937_64 rng(5);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> Ad((size_t)81 * K), Ae((size_t)81 * K), xs((size_t)9 * K);
  for (int p = 0; p < K; ++p) {
    double B[81];
    for (auto& v : B) v = 0.3 * nd(rng);
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) { double s = (r == c) ? 4.0 : 0.0; for (int k = 0; k < 9; ++k) s += B[9 * r + k] * B[9 * c + k]; Ad[(size_t)81 * p + 9 * r + c] = s; }
    for (int e = 0; e < 81; ++e) Ae[(size_t)81 * p + e] = 0.3 * nd(rng);
    for (int r = 0; r < 9; ++r) xs[(size_t)9 * p + r] = nd(rng);
  }
  std::vector<int> cp(NCH + 1);
  for (int c = 0; c <= NCH; ++c) cp[c] = c * NPOS;
  DevProblem P; std::memset(&P, 0, sizeof(P));
  P.K = K; P.vi = 1; P.nchains = NCH;
  auto up = [&](const void* h, size_t bytes) { void* d; hipMalloc(&d, bytes); if (h) hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); else hipMemset(d, 0, bytes); return d; };
  P.chain_ptr = (int*)up(cp.data(), cp.size() * 4);
  P.Ad = (double*)up(Ad.data(), Ad.size() * 8); P.Ae = (double*)up(Ae.data(), Ae.size() * 8);
  P.xs = (double*)up(xs.data(), xs.size() * 8);
  P.Ldinv = (double*)up(nullptr, Ad.size() * 8); P.Lsub = (double*)up(nullptr, Ad.size() * 8); P.zs = (double*)up(nullptr, xs.size() * 8);
  P.flag = (int*)up(nullptr, 16);
  P.Nback = (double*)up(nullptr, (size_t)90 * K * 8); P.Zfwd = (double*)up(nullptr, (size_t)90 * K * 8);
  std::vector<int> pce(K);
  for (int p = 0; p < K; ++p) pce[p] = (p / NPOS + 1) * NPOS;
  P.pos_chain_end = (int*)up(pce.data(), pce.size() * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nch : {1, 5}) {
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_sb_chain_factor, dim3(nch), dim3(64), 0, 0, P);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    printf("k_sb_chain_factor, %d chain(s) of %d positions: %.1f us = %.2f us per position\n", nch, NPOS, best * 1e3, best * 1e3 / NPOS);
  }
  // z by the forward sweep (xs holds b_s)
  hipLaunchKernelGGL(k_sb_sweep_mat, dim3((81 * K + 255) / 256), dim3(256), 0, 0, P);
  hipLaunchKernelGGL(k_sb_sweep_vec<1>, dim3((9 * K + 255) / 256), dim3(256), 0, 0, P, (const double*)P.xs);
  hipLaunchKernelGGL(k_sb_sweep<1>, dim3(NCH), dim3(64), 0, 0, P);
  hipDeviceSynchronize();
  // host recurrence on chain 0
  std::vector<double> Ld((size_t)81 * K), Ls((size_t)81 * K), zs((size_t)9 * K);
  hipMemcpy(Ld.data(), P.Ldinv, Ld.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(Ls.data(), P.Lsub, Ls.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(zs.data(), P.zs, zs.size() * 8, hipMemcpyDeviceToHost);
  double eLd = 0, eLs = 0, ez = 0;
  {
    double sub[81] = {0}, zprev[9] = {0};
    for (int p = 0; p < NPOS; ++p) {
      double M[81], L[81] = {0}, X[81] = {0};
      for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) { double m = Ad[(size_t)81 * p + 9 * a + b]; for (int k = 0; k < 9; ++k) m -= sub[9 * a + k] * sub[9 * b + k]; M[9 * a + b] = m; }
      for (int j = 0; j < 9; ++j) { double d = M[9 * j + j]; for (int k = 0; k < j; ++k) d -= L[9 * j + k] * L[9 * j + k]; d = std::sqrt(d); L[9 * j + j] = d;
        for (int i = j + 1; i < 9; ++i) { double s = M[9 * i + j]; for (int k = 0; k < j; ++k) s -= L[9 * i + k] * L[9 * j + k]; L[9 * i + j] = s / d; } }
      for (int c = 0; c < 9; ++c) for (int r = 0; r < 9; ++r) { double s = (r == c) ? 1.0 : 0.0; for (int k = 0; k < r; ++k) s -= L[9 * r + k] * X[9 * k + c]; X[9 * r + c] = s / L[9 * r + r]; }
      double v[9], z[9];
      for (int r = 0; r < 9; ++r) { v[r] = xs[(size_t)9 * p + r]; for (int k = 0; k < 9; ++k) v[r] -= sub[9 * r + k] * zprev[k]; }
      for (int r = 0; r < 9; ++r) { z[r] = 0; for (int k = 0; k <= r; ++k) z[r] += X[9 * r + k] * v[k]; ez = std::max(ez, std::fabs(z[r] - zs[(size_t)9 * p + r])); }
      for (int e = 0; e < 81; ++e) { eLd = std::max(eLd, std::fabs(X[e] - Ld[(size_t)81 * p + e])); eLs = std::max(eLs, std::fabs(sub[e] - Ls[(size_t)81 * p + e])); }
      if (p + 1 < NPOS) { double nx[81]; for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) { double s = 0; for (int c = 0; c < 9; ++c) s += Ae[(size_t)81 * (p + 1) + 9 * a + c] * X[9 * b + c]; nx[9 * a + b] = s; } std::memcpy(sub, nx, sizeof(sub)); }
      std::memcpy(zprev, z, sizeof(z));
    }
  }
  printf("  vs host recurrence (chain 0): max|Ldinv| err %.2e  max|Lsub| err %.2e  max|z| err %.2e\n", eLd, eLs, ez);
  for (int nch : {1, 5}) {
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_sb_backsolve, dim3(nch), dim3(64), 0, 0, P, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    printf("k_sb_backsolve (backward half), %d chain(s): %.1f us = %.2f us per position\n", nch, best * 1e3, best * 1e3 / NPOS);
  }
  // single-product form: same result?
  std::vector<double> xa((size_t)9 * K), xb((size_t)9 * K);
  hipMemcpy(P.xs, xs.data(), xs.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_sb_backsolve, dim3(NCH), dim3(64), 0, 0, P, 0);
  hipMemcpy(xa.data(), P.xs, xa.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(P.xs, xs.data(), xs.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_sb_sweep_mat, dim3((81 * K + 255) / 256), dim3(256), 0, 0, P);
  hipLaunchKernelGGL(k_sb_sweep_vec<-1>, dim3((9 * K + 255) / 256), dim3(256), 0, 0, P, (const double*)P.xs);
  hipLaunchKernelGGL(k_sb_sweep<-1>, dim3(NCH), dim3(64), 0, 0, P);
  hipMemcpy(xb.data(), P.xs, xb.size() * 8, hipMemcpyDeviceToHost);
  double ex = 0, xm = 0;
  for (size_t i = 0; i < xa.size(); ++i) { ex = std::max(ex, std::fabs(xa[i] - xb[i])); xm = std::max(xm, std::fabs(xa[i])); }
  printf("k_sb_sweep<-1> vs k_sb_backsolve: max diff %.2e (max|x| %.2e)\n", ex, xm);
  for (int nch : {1, 5}) {
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_sb_sweep<-1>, dim3(nch), dim3(64), 0, 0, P);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    printf("k_sb_sweep<-1> (one product per step), %d chain(s): %.1f us = %.3f us per position\n", nch, best * 1e3, best * 1e3 / NPOS);
  }
  return 0;
}
