// panel_probe.hip — correctness + phase timing of the 256-column panel chain kernels (k_panel.hip) on an idle GPU (dev tool):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCOVGPU_PROBE tools/panel_probe.hip -o /tmp/panel_probe && /tmp/panel_probe
// Checks k_potrf_panel (nb = 16 and 8, batched, with the right-hand side riding along), k_trsm_sub<16|8> and
// k_bwd_step_sub against a host Cholesky in double precision.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "../covins_amd/csrc/k_panel.hip"

using namespace covgpu;

static void host_chol(std::vector<double>& A, int n, int ld) {  // lower, in place
  for (int j = 0; j < n; ++j) {
    double d = A[j * ld + j];
    for (int k = 0; k < j; ++k) d -= A[j * ld + k] * A[j * ld + k];
    d = std::sqrt(d); A[j * ld + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * ld + j];
      for (int k = 0; k < j; ++k) s -= A[i * ld + k] * A[j * ld + k];
      A[i * ld + j] = s / d;
    }
  }
}

int main() {
  const int N = 1024, k0 = 256, NBT = 2;  // panel at (k0, k0) of an N-order matrix; rows below are TRSM'd
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd(0.0, 1.0);
  for (int nb : {16, 8}) {
    const int n = 16 * nb;
    std::vector<double> A((size_t)NBT * N * N, 0.0), b((size_t)NBT * 2 * N, 0.0);
    for (int bt = 0; bt < NBT; ++bt) {
      double* Ab = A.data() + (size_t)bt * N * N;
      // SPD diagonal block with a wide spectrum: B B^T + diag
      std::vector<double> Bm((size_t)n * n);
      for (auto& v : Bm) v = nd(rng);
      for (int r = 0; r < n; ++r)
        for (int c = 0; c <= r; ++c) {
          double s = 0;
          for (int k = 0; k < n; ++k) s += Bm[(size_t)r * n + k] * Bm[(size_t)c * n + k] * std::pow(10.0, -6.0 * k / n);
          Ab[(size_t)(k0 + r) * N + k0 + c] = s + (r == c ? 1e-3 : 0.0);
        }
      for (int r = 0; r < n; ++r)
        for (int c = r + 1; c < n; ++c) Ab[(size_t)(k0 + r) * N + k0 + c] = 777.0;  // the upper triangle must never be read
      for (int r = k0 + n; r < N; ++r)
        for (int c = 0; c < n; ++c) Ab[(size_t)r * N + k0 + c] = nd(rng);
      for (int r = 0; r < N; ++r) b[(size_t)bt * 2 * N + r] = nd(rng);
    }
    double *dA, *dD, *db; int* df;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dD, (size_t)NBT * (N / 128) * 128 * 128 * 8); hipMalloc(&db, b.size() * 8); hipMalloc(&df, 16);
    hipMemset(df, 0, 16);
    hipMemset(dD, 0, (size_t)NBT * (N / 128) * 128 * 128 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sM = (size_t)N * N, sL = (size_t)(N / 128) * 128 * 128, sR = (size_t)2 * N;
    const int t0 = k0 / 128, w = nb / 8;
    float best = 1e9; double ph[8] = {1e9, 1e9, 1e9, 1e9, 1e9, 1e9, 1e9, 1e9};
    for (int it = 0; it < 50; ++it) {
      hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
      hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice);
      { long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_pprobe), z, sizeof(z)); }
      hipEventRecord(e0);
      launch_potrf_panel(dA, N, t0, w, dD, df, db, N, 1, sM, sL, sR, 0);  // one workgroup: thread 0 of each would race on g_pprobe
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
      long long pr[8]; hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_pprobe), sizeof(pr));
      for (int k = 0; k <= 7; ++k) ph[k] = std::min(ph[k], pr[k] / 2400.0);  // shader cycles at 2.4 GHz
    }
    for (int nbt : {1, 2, 5}) {  // is the batch concurrent? (needs NBT >= nbt matrices: reuse the first)
      float bq = 1e9;
      for (int it = 0; it < 20; ++it) {
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        launch_potrf_panel(dA, N, t0, w, dD, df, db, N, nbt, nbt <= NBT ? sM : 0, nbt <= NBT ? sL : 0, nbt <= NBT ? sR : 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); bq = std::min(bq, ms);
      }
      printf("  launch with %d workgroup(s): %.1f us\n", nbt, bq * 1e3);
    }
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice);
    launch_potrf_panel(dA, N, t0, w, dD, df, db, N, NBT, sM, sL, sR, 0);
    hipDeviceSynchronize();
    printf("k_potrf_panel nb=%d (x%d batched): best %.1f us; phases (sum over %d steps) A %.1f (wave 0's diagonal block alone %.1f)  B %.1f  C %.1f us; prologue %.1f us; wave 1: tile updates %.1f  L store + rhs %.1f us\n",
           nb, NBT, best * 1e3, nb, ph[1], ph[5], ph[2], ph[3], ph[4], ph[6], ph[7]);
    if (nb == 16) {
      long long ps[4][16]; hipMemcpyFromSymbol(ps, HIP_SYMBOL(g_pstep), sizeof(ps));
      printf("  step: sweep done | past X | past Y | wave %d updates done  [us from kernel start]\n", PPROBE_WAVE);
      for (int j = 0; j < 16; ++j) printf("   %2d: %6.2f %6.2f %6.2f %6.2f\n", j, ps[0][j] / 2400.0, ps[1][j] / 2400.0, ps[2][j] / 2400.0, ps[3][j] / 2400.0);
    }
    if (nb == 16) {
      long long pa[16][16]; hipMemcpyFromSymbol(pa, HIP_SYMBOL(g_parr), sizeof(pa));
      printf("  arrival at X(j) [us], waves 1..15:\n");
      for (int j = 0; j < 16; ++j) { printf("   %2d:", j); for (int w = 1; w < 16; ++w) printf(" %6.2f", pa[w][j] / 2400.0); printf("\n"); }
    }
    // rows below by block substitution
    float bestT = 1e9;
    std::vector<double> Apost(A.size());
    hipMemcpy(Apost.data(), dA, A.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> bpost(b.size());
    hipMemcpy(bpost.data(), db, b.size() * 8, hipMemcpyDeviceToHost);
    for (int it = 0; it < 20; ++it) {
      hipMemcpy(dA, Apost.data(), A.size() * 8, hipMemcpyHostToDevice);
      hipMemcpy(db, bpost.data(), b.size() * 8, hipMemcpyHostToDevice);
      hipEventRecord(e0);
      launch_trsm_sub(dA, N, t0, w, t0 + w, N / 128, dD, db, N, NBT, sM, sL, sR, nullptr, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); bestT = std::min(bestT, ms);
    }
    float warmT = 1e9;   // the same launch on data the chip has just touched (the numbers are garbage the second time: timing only)
    for (int it = 0; it < 10; ++it) {
      hipMemcpy(dA, Apost.data(), A.size() * 8, hipMemcpyHostToDevice);
      hipMemcpy(db, bpost.data(), b.size() * 8, hipMemcpyHostToDevice);
      launch_trsm_sub(dA, N, t0, w, t0 + w, N / 128, dD, db, N, NBT, sM, sL, sR, nullptr, 0, 0);
      hipEventRecord(e0);
      launch_trsm_sub(dA, N, t0, w, t0 + w, N / 128, dD, db, N, NBT, sM, sL, sR, nullptr, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); warmT = std::min(warmT, ms);
    }
    hipMemcpy(dA, Apost.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, bpost.data(), b.size() * 8, hipMemcpyHostToDevice);
    launch_trsm_sub(dA, N, t0, w, t0 + w, N / 128, dD, db, N, NBT, sM, sL, sR, nullptr, 0, 0);
    printf("  (second of two back-to-back launches: %.1f us)\n", warmT * 1e3);
    if (nb == 16) {
      long long tp[4][24]; hipMemcpyFromSymbol(tp, HIP_SYMBOL(g_tprobe), sizeof(tp));
      printf("  four-wave substitution, slab 0: Z_j in registers at [us] (rows: waves 0-3; last column: all updates done)\n");
      for (int wv = 0; wv < 4; ++wv) { printf("   "); for (int k = 0; k <= 16; ++k) printf(" %5.2f", tp[wv][k] / 2400.0); printf("\n"); }
    }
    std::vector<double> G(A.size()), D((size_t)NBT * sL), bg(b.size());
    hipMemcpy(G.data(), dA, A.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(bg.data(), db, b.size() * 8, hipMemcpyDeviceToHost);
    int fl = 0; hipMemcpy(&fl, df, 4, hipMemcpyDeviceToHost);
    double eL = 0, eY = 0, eD = 0, eX = 0, eR = 0, eU = 0, lmax = 0;
    for (int bt = 0; bt < NBT; ++bt) {
      std::vector<double> H((size_t)n * n);
      const double* Ab = A.data() + (size_t)bt * N * N;
      const double* Gb = G.data() + (size_t)bt * N * N;
      for (int r = 0; r < n; ++r) for (int c = 0; c <= r; ++c) H[(size_t)r * n + c] = Ab[(size_t)(k0 + r) * N + k0 + c];
      host_chol(H, n, n);
      for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
          const double g = Gb[(size_t)(k0 + r) * N + k0 + c];
          if (c <= r) { eL = std::max(eL, std::fabs(g - H[(size_t)r * n + c])); lmax = std::max(lmax, std::fabs(H[(size_t)r * n + c])); }
          else eU = std::max(eU, std::fabs(g - 777.0));
        }
      // y = L^-1 b[k0 .. k0+n)
      std::vector<double> y(n);
      for (int r = 0; r < n; ++r) {
        double s = b[(size_t)bt * 2 * N + k0 + r];
        for (int k = 0; k < r; ++k) s -= H[(size_t)r * n + k] * y[k];
        y[r] = s / H[(size_t)r * n + r];
        eY = std::max(eY, std::fabs(y[r] - bg[(size_t)bt * 2 * N + N + k0 + r]));
      }
      // Dinv_j L_jj = I
      for (int j = 0; j < nb; ++j) {
        const double* Dj = D.data() + (size_t)bt * sL + (size_t)(t0 + (j >> 3)) * 128 * 128 + (size_t)(j & 7) * 256;
        for (int r = 0; r < 16; ++r)
          for (int c = 0; c < 16; ++c) {
            double s = 0;
            for (int k = 0; k < 16; ++k) s += Dj[r * 16 + k] * (k >= c ? H[(size_t)(16 * j + k) * n + 16 * j + c] : 0.0);
            eD = std::max(eD, std::fabs(s - (r == c ? 1.0 : 0.0)));
          }
      }
      // X = A L^-T for the rows below, rhs[rows] -= X y
      for (int r = k0 + n; r < N; ++r) {
        std::vector<double> x(n);
        for (int c = 0; c < n; ++c) {
          double s = Ab[(size_t)r * N + k0 + c];
          for (int k = 0; k < c; ++k) s -= x[k] * H[(size_t)c * n + k];
          x[c] = s / H[(size_t)c * n + c];
          eX = std::max(eX, std::fabs(x[c] - Gb[(size_t)r * N + k0 + c]));
        }
        double s = b[(size_t)bt * 2 * N + r];
        for (int c = 0; c < n; ++c) s -= x[c] * y[c];
        eR = std::max(eR, std::fabs(s - bg[(size_t)bt * 2 * N + r]));
      }
    }
    printf("  max|L - L_host| %.3e (max|L| %.2e)  max|y - y_host| %.3e  max|Dinv L - I| %.3e  upper triangle touched %.1e  flag %d\n", eL, lmax, eY, eD, eU, fl);
    printf("k_trsm_sub<%d> on %d slabs x%d: best %.1f us   max|X - X_host| %.3e  max|rhs - rhs_host| %.3e\n", nb, (N - k0 - n) / 16, NBT, bestT * 1e3, eX, eR);
    if (nb == 8) {
      // backward step on tile t0: x_p = L_pp^-T y_p, y[cols < k0] -= L[tile rows, cols]^T x_p (the columns left of the tile
      // hold arbitrary numbers here: fill them)
      std::vector<double> Lfull(G);
      for (int bt = 0; bt < NBT; ++bt)
        for (int r = k0; r < k0 + 128; ++r)
          for (int c = 0; c < k0; ++c) Lfull[(size_t)bt * N * N + (size_t)r * N + c] = nd(rng);
      std::vector<double> yv(b.size());
      for (auto& v : yv) v = nd(rng);
      hipMemcpy(dA, Lfull.data(), A.size() * 8, hipMemcpyHostToDevice);
      hipMemcpy(db, yv.data(), b.size() * 8, hipMemcpyHostToDevice);
      float bestB = 1e9;
      for (int it = 0; it < 20; ++it) {
        hipMemcpy(db, yv.data(), b.size() * 8, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        launch_bwd_step_sub(dA, N, t0, dD + (size_t)t0 * 128 * 128, db + N, db, k0, (k0 + 31) / 32, NBT, sM, sL, sR, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); bestB = std::min(bestB, ms);
      }
      std::vector<double> out(b.size());
      hipMemcpy(out.data(), db, b.size() * 8, hipMemcpyDeviceToHost);
      double eXp = 0, eYl = 0;
      for (int bt = 0; bt < NBT; ++bt) {
        const double* Lb = Lfull.data() + (size_t)bt * N * N;
        std::vector<double> xp(128);
        for (int r = 127; r >= 0; --r) {
          double s = yv[(size_t)bt * 2 * N + N + k0 + r];
          for (int k = r + 1; k < 128; ++k) s -= Lb[(size_t)(k0 + k) * N + k0 + r] * xp[k];
          xp[r] = s / Lb[(size_t)(k0 + r) * N + k0 + r];
          eXp = std::max(eXp, std::fabs(xp[r] - out[(size_t)bt * 2 * N + k0 + r]));
        }
        for (int c = 0; c < k0; ++c) {
          double s = yv[(size_t)bt * 2 * N + N + c];
          for (int r = 0; r < 128; ++r) s -= Lb[(size_t)(k0 + r) * N + c] * xp[r];
          eYl = std::max(eYl, std::fabs(s - out[(size_t)bt * 2 * N + N + c]));
        }
      }
      printf("k_bwd_step_sub: best %.1f us   max|x_p - host| %.3e  max|y_left - host| %.3e\n", bestB * 1e3, eXp, eYl);
    }
    hipFree(dA); hipFree(dD); hipFree(db); hipFree(df);
  }
  return 0;
}
