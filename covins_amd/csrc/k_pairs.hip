// k_pairs.hip — covisible keyframe pairs on the device: the static structure of the landmark elimination
// (which pose-pose blocks S = H_pp - sum_l W H_ll^-1 W^T fills, A.6) and the covisibility weights of
// Keyframe::UpdateCovisibilityConnections (keyframe_be.cpp:559-608).
//
// Both are "for every landmark, every pair of its observing keyframes": sum n_l^2 entries (8.7 M on the 5-agent map). The
// host version (counting pass, fill pass, per-row stable sorts on 16 threads) was the longest stage of an upload (28 of
// 44 ms). Here: one thread per landmark counts and emits 64-bit keys (row keyframe << 32 | column keyframe) with the pair
// of observation indices as value, ONE stable radix sort (rocPRIM's device primitives, called directly — setup code, not a hot kernel),
// run-length encoding of the sorted keys = the unique pairs with their common-landmark counts. Stable sort + emission in
// landmark order = the fixed summation order k_pair_blocks relies on (bit-reproducibility contract, k_visual.hip).
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>

#include "common.hpp"

namespace covgpu {

// key_of_kf[kf] >= 0: the keyframe's index in the pair numbering (chain position, or the IR index itself); < 0: left out
__global__ __launch_bounds__(256) void k_pair_count(int L, const int* __restrict__ lm_obs_ptr, const int* __restrict__ obs_kf, const int* __restrict__ key_of_kf,
                                                     unsigned long long* __restrict__ cnt) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int o0 = lm_obs_ptr[l], o1 = lm_obs_ptr[l + 1];
  unsigned long long n = 0;
  for (int a = o0; a < o1; ++a) {
    const int pa = key_of_kf[obs_kf[a]];
    if (pa < 0) continue;
    for (int b = o0; b < o1; ++b) { const int pb = key_of_kf[obs_kf[b]]; n += (pb >= 0 && pb < pa) ? 1 : 0; }
  }
  cnt[l] = n;
}
__global__ __launch_bounds__(256) void k_pair_emit(int L, const int* __restrict__ lm_obs_ptr, const int* __restrict__ obs_kf, const int* __restrict__ key_of_kf,
                                                    const unsigned long long* __restrict__ off, unsigned long long* __restrict__ keys,
                                                    unsigned long long* __restrict__ vals) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int o0 = lm_obs_ptr[l], o1 = lm_obs_ptr[l + 1];
  unsigned long long e = off[l];
  for (int a = o0; a < o1; ++a) {
    const int pa = key_of_kf[obs_kf[a]];
    if (pa < 0) continue;
    for (int b = o0; b < o1; ++b) {
      const int pb = key_of_kf[obs_kf[b]];
      if (pb >= 0 && pb < pa) { keys[e] = ((unsigned long long)pa << 32) | (unsigned)pb; vals[e] = ((unsigned long long)a << 32) | (unsigned)b; ++e; }
    }
  }
}
__global__ __launch_bounds__(256) void k_pair_split(size_t n, const unsigned long long* __restrict__ src, int* __restrict__ hi, int* __restrict__ lo) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q < n) { hi[q] = (int)(src[q] >> 32); lo[q] = (int)(src[q] & 0xffffffffull); }
}
__global__ __launch_bounds__(256) void k_pair_ptr(int n, const unsigned long long* __restrict__ scan, const int* __restrict__ counts, int* __restrict__ ptr) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < n) ptr[q] = (int)scan[q];
  if (q == n - 1) ptr[n] = (int)(scan[q] + (unsigned long long)counts[q]);
}
__global__ __launch_bounds__(256) void k_count_widen(int n, const int* __restrict__ c, unsigned long long* __restrict__ w) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < n) w[q] = (unsigned long long)c[q];
}

// ------------------------------------------------------------------------------------------------ second round of a GBA call, on the device
// optimization_be.cpp:296-557 rebuilds the ceres::Problem after the outlier round: the erased observations are gone, landmarks left with
// fewer than two observations are skipped (:428-440), everything else is the first round's problem. Here the observation stream of the
// RESIDENT first-round problem is compacted in place of a second Map -> IR flatten + upload: keep flags, two exclusive scans, one
// gather, and the keyframe-major lists by ONE stable radix sort (keyframe as key, new observation index as value: ascending
// observation order inside a keyframe, as the host loop of an upload produces it).
__global__ __launch_bounds__(256) void k_r2_keep_lm(int L, const int* __restrict__ left, unsigned* __restrict__ keepL) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l < L) keepL[l] = left[l] >= 2 ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_r2_keep_obs(int O, const unsigned char* __restrict__ erase, const int* __restrict__ obs_lm, const unsigned* __restrict__ keepL,
                                                      unsigned* __restrict__ keepO) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o < O) keepO[o] = (!erase[o] && keepL[obs_lm[o]]) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_r2_gather_lm(int L, const unsigned* __restrict__ keepL, const unsigned* __restrict__ lpos, const int* __restrict__ left,
                                                       const double* __restrict__ lm0, double* __restrict__ lm0n, int* __restrict__ nobs, int* __restrict__ lm_old) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= L || !keepL[l]) return;
  const unsigned q = lpos[l];
  lm0n[3 * (size_t)q] = lm0[3 * (size_t)l]; lm0n[3 * (size_t)q + 1] = lm0[3 * (size_t)l + 1]; lm0n[3 * (size_t)q + 2] = lm0[3 * (size_t)l + 2];
  nobs[q] = left[l];   // (every observation a kept landmark keeps is a kept observation)
  lm_old[q] = l;
}
__global__ __launch_bounds__(256) void k_r2_gather_obs(int O, const unsigned* __restrict__ keepO, const unsigned* __restrict__ opos, const unsigned* __restrict__ lpos,
                                                        const int* __restrict__ obs_kf, const int* __restrict__ obs_lm, const double* __restrict__ u,
                                                        const double* __restrict__ v, const double* __restrict__ sg, int* __restrict__ obs_kf_n,
                                                        int* __restrict__ obs_lm_n, double* __restrict__ un, double* __restrict__ vn, double* __restrict__ sgn,
                                                        int* __restrict__ iota) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= O || !keepO[o]) return;
  const unsigned q = opos[o];
  obs_kf_n[q] = obs_kf[o]; obs_lm_n[q] = (int)lpos[obs_lm[o]]; un[q] = u[o]; vn[q] = v[o]; sgn[q] = sg[o]; iota[q] = (int)q;
}
__global__ __launch_bounds__(256) void k_r2_kf_count(int O, const int* __restrict__ obs_kf, unsigned* __restrict__ cnt) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o < O) atomicAdd(&cnt[obs_kf[o]], 1u);   // (integer counts: order-independent)
}
__global__ __launch_bounds__(256) void k_r2_ptr(int n, const unsigned* __restrict__ scan, const unsigned* __restrict__ cnt, int* __restrict__ ptr) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < n) ptr[q] = (int)scan[q];
  if (q == n - 1) ptr[n] = (int)(scan[q] + cnt[q]);
}
__global__ __launch_bounds__(256) void k_r2_widen(int n, const int* __restrict__ c, unsigned* __restrict__ w) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < n) w[q] = (unsigned)c[q];
}

#define R2_HIP(expr) do { if ((expr) != hipSuccess) { ok = false; goto done; } } while (0)
// out.* are hipMalloc'ed here and owned by the caller. L2 / O2: landmarks / observations of the second round.
bool round2_compact_device(int L, int O, int K, const unsigned char* d_erase, const int* d_left, const int* d_obs_kf, const int* d_obs_lm, const double* d_u,
                           const double* d_v, const double* d_sigma, const double* d_lm0, hipStream_t st, Round2Lists& out) {
  out = Round2Lists();
  bool ok = true;
  unsigned *keepL = nullptr, *keepO = nullptr, *lpos = nullptr, *opos = nullptr, *kcnt = nullptr, *kscan = nullptr, *nw = nullptr, *nscan = nullptr;
  int *nobs = nullptr, *iota = nullptr, *keys2 = nullptr;
  void* tmp = nullptr; size_t tmp_bytes = 0;
  unsigned tail[4] = {0, 0, 0, 0};
  int kbits = 1;
  while ((1 << kbits) < K) ++kbits;
  auto need_tmp = [&](size_t bytes) { if (bytes > tmp_bytes) { if (tmp) (void)hipFree(tmp); tmp = nullptr; if (hipMalloc(&tmp, bytes) != hipSuccess) return false; tmp_bytes = bytes; } return true; };
  auto scan = [&](unsigned* in, unsigned* o2, size_t n) {
    size_t b = 0;
    if (rocprim::exclusive_scan(nullptr, b, in, o2, 0u, n, rocprim::plus<unsigned>(), st) != hipSuccess) return false;
    if (!need_tmp(b)) return false;
    return rocprim::exclusive_scan(tmp, b, in, o2, 0u, n, rocprim::plus<unsigned>(), st) == hipSuccess;
  };
  if (L <= 0 || O <= 0) return true;   // nothing to compact: L2 = O2 = 0
  R2_HIP(hipMalloc((void**)&keepL, (size_t)L * 4)); R2_HIP(hipMalloc((void**)&lpos, (size_t)L * 4));
  R2_HIP(hipMalloc((void**)&keepO, (size_t)O * 4)); R2_HIP(hipMalloc((void**)&opos, (size_t)O * 4));
  hipLaunchKernelGGL(k_r2_keep_lm, dim3((L + 255) / 256), dim3(256), 0, st, L, d_left, keepL);
  hipLaunchKernelGGL(k_r2_keep_obs, dim3((O + 255) / 256), dim3(256), 0, st, O, d_erase, d_obs_lm, (const unsigned*)keepL, keepO);
  if (!scan(keepL, lpos, (size_t)L) || !scan(keepO, opos, (size_t)O)) { ok = false; goto done; }
  R2_HIP(hipMemcpyAsync(&tail[0], keepL + (L - 1), 4, hipMemcpyDeviceToHost, st)); R2_HIP(hipMemcpyAsync(&tail[1], lpos + (L - 1), 4, hipMemcpyDeviceToHost, st));
  R2_HIP(hipMemcpyAsync(&tail[2], keepO + (O - 1), 4, hipMemcpyDeviceToHost, st)); R2_HIP(hipMemcpyAsync(&tail[3], opos + (O - 1), 4, hipMemcpyDeviceToHost, st));
  R2_HIP(hipStreamSynchronize(st));
  out.L2 = (int)(tail[0] + tail[1]); out.O2 = (int)(tail[2] + tail[3]);
  {
    const size_t L2 = (size_t)std::max(out.L2, 1), O2 = (size_t)std::max(out.O2, 1);
    R2_HIP(hipMalloc((void**)&out.lm0, 3 * L2 * 8)); R2_HIP(hipMalloc((void**)&out.lm_obs_ptr, (L2 + 1) * 4)); R2_HIP(hipMalloc((void**)&out.lm_old, L2 * 4));
    R2_HIP(hipMalloc((void**)&out.obs_kf, O2 * 4)); R2_HIP(hipMalloc((void**)&out.obs_lm, O2 * 4));
    R2_HIP(hipMalloc((void**)&out.obs_u, O2 * 8)); R2_HIP(hipMalloc((void**)&out.obs_v, O2 * 8)); R2_HIP(hipMalloc((void**)&out.obs_sigma, O2 * 8));
    R2_HIP(hipMalloc((void**)&out.kf_obs_ptr, ((size_t)K + 1) * 4)); R2_HIP(hipMalloc((void**)&out.kf_obs_idx, O2 * 4));
    R2_HIP(hipMalloc((void**)&nobs, L2 * 4)); R2_HIP(hipMalloc((void**)&nw, L2 * 4)); R2_HIP(hipMalloc((void**)&nscan, L2 * 4));
    R2_HIP(hipMalloc((void**)&iota, O2 * 4)); R2_HIP(hipMalloc((void**)&keys2, O2 * 4));
    R2_HIP(hipMalloc((void**)&kcnt, (size_t)K * 4)); R2_HIP(hipMalloc((void**)&kscan, (size_t)K * 4));
    R2_HIP(hipMemsetAsync(kcnt, 0, (size_t)K * 4, st));
    R2_HIP(hipMemsetAsync(out.lm_obs_ptr, 0, (L2 + 1) * 4, st)); R2_HIP(hipMemsetAsync(out.kf_obs_ptr, 0, ((size_t)K + 1) * 4, st));
  }
  if (out.L2 > 0 && out.O2 > 0) {
    hipLaunchKernelGGL(k_r2_gather_lm, dim3((L + 255) / 256), dim3(256), 0, st, L, (const unsigned*)keepL, (const unsigned*)lpos, d_left, d_lm0, out.lm0, nobs, out.lm_old);
    hipLaunchKernelGGL(k_r2_gather_obs, dim3((O + 255) / 256), dim3(256), 0, st, O, (const unsigned*)keepO, (const unsigned*)opos, (const unsigned*)lpos, d_obs_kf, d_obs_lm,
                       d_u, d_v, d_sigma, out.obs_kf, out.obs_lm, out.obs_u, out.obs_v, out.obs_sigma, iota);
    // landmark-major pointers of the kept landmarks
    hipLaunchKernelGGL(k_r2_widen, dim3((out.L2 + 255) / 256), dim3(256), 0, st, out.L2, (const int*)nobs, nw);
    if (!scan(nw, nscan, (size_t)out.L2)) { ok = false; goto done; }
    hipLaunchKernelGGL(k_r2_ptr, dim3((out.L2 + 255) / 256), dim3(256), 0, st, out.L2, (const unsigned*)nscan, (const unsigned*)nw, out.lm_obs_ptr);
    // keyframe-major lists
    hipLaunchKernelGGL(k_r2_kf_count, dim3((out.O2 + 255) / 256), dim3(256), 0, st, out.O2, (const int*)out.obs_kf, kcnt);
    if (!scan(kcnt, kscan, (size_t)K)) { ok = false; goto done; }
    hipLaunchKernelGGL(k_r2_ptr, dim3((K + 255) / 256), dim3(256), 0, st, K, (const unsigned*)kscan, (const unsigned*)kcnt, out.kf_obs_ptr);
    {
      size_t b = 0;
      R2_HIP(rocprim::radix_sort_pairs(nullptr, b, out.obs_kf, keys2, iota, out.kf_obs_idx, (size_t)out.O2, 0u, (unsigned)kbits, st));
      if (!need_tmp(b)) { ok = false; goto done; }
      R2_HIP(rocprim::radix_sort_pairs(tmp, b, out.obs_kf, keys2, iota, out.kf_obs_idx, (size_t)out.O2, 0u, (unsigned)kbits, st));
    }
  }
  R2_HIP(hipStreamSynchronize(st));
done:
  for (void* q : {(void*)keepL, (void*)keepO, (void*)lpos, (void*)opos, (void*)kcnt, (void*)kscan, (void*)nw, (void*)nscan, (void*)nobs, (void*)iota, (void*)keys2, tmp}) if (q) (void)hipFree(q);
  if (!ok) { out.free_all(); out = Round2Lists(); }
  return ok;
}

// ------------------------------------------------------------------------------------------------ upload helpers (were host loops over O)
// interleaved keypoints [O][2] -> obs_u / obs_v, landmark index of every observation from lm_obs_ptr, identity permutation
__global__ __launch_bounds__(256) void k_obs_unpack(int L, int O, const int* __restrict__ lm_obs_ptr, const double* __restrict__ uv, int* __restrict__ obs_lm,
                                                     double* __restrict__ u, double* __restrict__ v, int* __restrict__ iota) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < O) { u[t] = uv[2 * (size_t)t]; v[t] = uv[2 * (size_t)t + 1]; iota[t] = t; }
  if (t < L) for (int o = lm_obs_ptr[t]; o < lm_obs_ptr[t + 1]; ++o) obs_lm[o] = t;
}
// keyframe-major observation lists: kf_obs_ptr [K + 1], kf_obs_idx [O] = the observations of every keyframe in ascending observation
// order (count, exclusive scan, ONE stable radix sort with the keyframe as key). d_iota: 0 .. O-1. Outputs are caller-allocated.
bool build_kf_lists_device(int O, int K, const int* d_obs_kf, const int* d_iota, int* kf_obs_ptr, int* kf_obs_idx, hipStream_t st) {
  bool ok = true;
  unsigned *kcnt = nullptr, *kscan = nullptr; int* keys2 = nullptr; void* tmp = nullptr;
  int kbits = 1;
  while ((1 << kbits) < K) ++kbits;
  if (hipMemsetAsync(kf_obs_ptr, 0, ((size_t)K + 1) * 4, st) != hipSuccess) return false;
  if (O <= 0) return true;
  if (hipMalloc((void**)&kcnt, (size_t)K * 4) != hipSuccess || hipMalloc((void**)&kscan, (size_t)K * 4) != hipSuccess || hipMalloc((void**)&keys2, (size_t)O * 4) != hipSuccess) ok = false;
  if (ok) {
    (void)hipMemsetAsync(kcnt, 0, (size_t)K * 4, st);
    hipLaunchKernelGGL(k_r2_kf_count, dim3((O + 255) / 256), dim3(256), 0, st, O, d_obs_kf, kcnt);
    size_t b1 = 0, b2 = 0;
    ok = rocprim::exclusive_scan(nullptr, b1, kcnt, kscan, 0u, (size_t)K, rocprim::plus<unsigned>(), st) == hipSuccess &&
         rocprim::radix_sort_pairs(nullptr, b2, d_obs_kf, keys2, d_iota, kf_obs_idx, (size_t)O, 0u, (unsigned)kbits, st) == hipSuccess;
    const size_t b = std::max(b1, b2);
    if (ok && hipMalloc(&tmp, std::max<size_t>(b, 16)) != hipSuccess) ok = false;
    if (ok) {
      size_t bb = b1;
      ok = rocprim::exclusive_scan(tmp, bb, kcnt, kscan, 0u, (size_t)K, rocprim::plus<unsigned>(), st) == hipSuccess;
      hipLaunchKernelGGL(k_r2_ptr, dim3((K + 255) / 256), dim3(256), 0, st, K, (const unsigned*)kscan, (const unsigned*)kcnt, kf_obs_ptr);
      bb = b2;
      ok = ok && rocprim::radix_sort_pairs(tmp, bb, d_obs_kf, keys2, d_iota, kf_obs_idx, (size_t)O, 0u, (unsigned)kbits, st) == hipSuccess;
      ok = ok && hipStreamSynchronize(st) == hipSuccess;
    }
  }
  for (void* q : {(void*)kcnt, (void*)kscan, (void*)keys2, tmp}) if (q) (void)hipFree(q);
  return ok;
}
void launch_obs_unpack(int L, int O, const int* lm_obs_ptr, const double* uv, int* obs_lm, double* u, double* v, int* iota, hipStream_t st) {
  const int n = std::max(L, O);
  if (n > 0) hipLaunchKernelGGL(k_obs_unpack, dim3((n + 255) / 256), dim3(256), 0, st, L, O, lm_obs_ptr, uv, obs_lm, u, v, iota);
}

#define PB_HIP(expr) do { if ((expr) != hipSuccess) { ok = false; goto done; } } while (0)

// Builds the pair structure. Outputs are hipMalloc'ed here (the caller owns them: out.* and frees with hipFree, or hands them
// to its allocation list). want_obs: also the per-entry observation indices (oa, ob). Returns false on an allocation failure.
bool build_pairs_device(int L, int K, const int* d_lm_obs_ptr, const int* d_obs_kf, const int* d_key_of_kf, bool want_obs, hipStream_t st, PairLists& out) {
  out = PairLists();
  bool ok = true;
  unsigned long long *d_cnt = nullptr, *d_off = nullptr, *d_keys = nullptr, *d_vals = nullptr, *d_keys2 = nullptr, *d_vals2 = nullptr, *d_uniq = nullptr, *d_scan = nullptr;
  int *d_counts = nullptr, *d_nruns = nullptr;
  void* d_tmp = nullptr; size_t tmp_bytes = 0;
  unsigned long long total = 0, last_cnt = 0, last_off = 0;
  int nruns = 0;
  int kbits = 1;
  while ((1 << kbits) < K) ++kbits;
  auto need_tmp = [&](size_t bytes) { if (bytes > tmp_bytes) { if (d_tmp) (void)hipFree(d_tmp); d_tmp = nullptr; if (hipMalloc(&d_tmp, bytes) != hipSuccess) return false; tmp_bytes = bytes; } return true; };
  if (L <= 0) { goto empty; }
  PB_HIP(hipMalloc((void**)&d_cnt, (size_t)L * 8)); PB_HIP(hipMalloc((void**)&d_off, (size_t)L * 8));
  hipLaunchKernelGGL(k_pair_count, dim3((L + 255) / 256), dim3(256), 0, st, L, d_lm_obs_ptr, d_obs_kf, d_key_of_kf, d_cnt);
  {
    size_t b = 0;
    PB_HIP(rocprim::exclusive_scan(nullptr, b, d_cnt, d_off, 0ull, (size_t)L, rocprim::plus<unsigned long long>(), st));
    if (!need_tmp(b)) { ok = false; goto done; }
    PB_HIP(rocprim::exclusive_scan(d_tmp, b, d_cnt, d_off, 0ull, (size_t)L, rocprim::plus<unsigned long long>(), st));
  }
  PB_HIP(hipMemcpyAsync(&last_cnt, d_cnt + (L - 1), 8, hipMemcpyDeviceToHost, st));
  PB_HIP(hipMemcpyAsync(&last_off, d_off + (L - 1), 8, hipMemcpyDeviceToHost, st));
  PB_HIP(hipStreamSynchronize(st));
  total = last_cnt + last_off;
  if (total == 0) goto empty;
  if (total >= (1ull << 31)) { ok = false; goto done; }  // (entry indices are 32-bit)
  PB_HIP(hipMalloc((void**)&d_keys, total * 8)); PB_HIP(hipMalloc((void**)&d_vals, total * 8));
  PB_HIP(hipMalloc((void**)&d_keys2, total * 8)); PB_HIP(hipMalloc((void**)&d_vals2, total * 8));
  hipLaunchKernelGGL(k_pair_emit, dim3((L + 255) / 256), dim3(256), 0, st, L, d_lm_obs_ptr, d_obs_kf, d_key_of_kf, d_off, d_keys, d_vals);
  {
    size_t b = 0;
    PB_HIP(rocprim::radix_sort_pairs(nullptr, b, d_keys, d_keys2, d_vals, d_vals2, (size_t)total, 0u, (unsigned)(32 + kbits), st));
    if (!need_tmp(b)) { ok = false; goto done; }
    PB_HIP(rocprim::radix_sort_pairs(d_tmp, b, d_keys, d_keys2, d_vals, d_vals2, (size_t)total, 0u, (unsigned)(32 + kbits), st));
  }
  // unique pairs + number of common landmarks of each (d_keys / d_vals are free again: reused as outputs)
  PB_HIP(hipMalloc((void**)&d_counts, total * 4)); PB_HIP(hipMalloc((void**)&d_nruns, 4));
  d_uniq = d_keys;
  {
    size_t b = 0;
    PB_HIP(rocprim::run_length_encode(nullptr, b, d_keys2, (unsigned)total, d_uniq, d_counts, d_nruns, st));
    if (!need_tmp(b)) { ok = false; goto done; }
    PB_HIP(rocprim::run_length_encode(d_tmp, b, d_keys2, (unsigned)total, d_uniq, d_counts, d_nruns, st));
  }
  PB_HIP(hipMemcpyAsync(&nruns, d_nruns, 4, hipMemcpyDeviceToHost, st));
  PB_HIP(hipStreamSynchronize(st));
  out.npairs = nruns; out.nent = (size_t)total;
  PB_HIP(hipMalloc((void**)&out.pair_i, (size_t)nruns * 4)); PB_HIP(hipMalloc((void**)&out.pair_j, (size_t)nruns * 4));
  PB_HIP(hipMalloc((void**)&out.pair_ptr, ((size_t)nruns + 1) * 4));
  hipLaunchKernelGGL(k_pair_split, dim3((unsigned)((nruns + 255) / 256)), dim3(256), 0, st, (size_t)nruns, d_uniq, out.pair_i, out.pair_j);
  d_scan = d_vals;  // widened counts + their scan share the freed value buffer (2 nruns <= total entries... not guaranteed: own buffers below if short)
  if ((size_t)2 * nruns > total) { d_scan = nullptr; PB_HIP(hipMalloc((void**)&d_scan, (size_t)2 * nruns * 8)); }
  hipLaunchKernelGGL(k_count_widen, dim3((nruns + 255) / 256), dim3(256), 0, st, nruns, d_counts, d_scan);
  {
    size_t b = 0;
    PB_HIP(rocprim::exclusive_scan(nullptr, b, d_scan, d_scan + nruns, 0ull, (size_t)nruns, rocprim::plus<unsigned long long>(), st));
    if (!need_tmp(b)) { ok = false; goto done; }
    PB_HIP(rocprim::exclusive_scan(d_tmp, b, d_scan, d_scan + nruns, 0ull, (size_t)nruns, rocprim::plus<unsigned long long>(), st));
  }
  hipLaunchKernelGGL(k_pair_ptr, dim3((nruns + 255) / 256), dim3(256), 0, st, nruns, d_scan + nruns, d_counts, out.pair_ptr);
  if (want_obs) {
    PB_HIP(hipMalloc((void**)&out.pair_oa, (size_t)total * 4)); PB_HIP(hipMalloc((void**)&out.pair_ob, (size_t)total * 4));
    hipLaunchKernelGGL(k_pair_split, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (size_t)total, d_vals2, out.pair_oa, out.pair_ob);
  }
  PB_HIP(hipStreamSynchronize(st));
  goto done;
empty:
  PB_HIP(hipMalloc((void**)&out.pair_ptr, 4)); PB_HIP(hipMemsetAsync(out.pair_ptr, 0, 4, st));
  PB_HIP(hipMalloc((void**)&out.pair_i, 4)); PB_HIP(hipMalloc((void**)&out.pair_j, 4));
  if (want_obs) { PB_HIP(hipMalloc((void**)&out.pair_oa, 4)); PB_HIP(hipMalloc((void**)&out.pair_ob, 4)); }
  PB_HIP(hipStreamSynchronize(st));
done:
  if (d_scan != nullptr && d_scan != d_vals) (void)hipFree(d_scan);
  for (void* q : {(void*)d_cnt, (void*)d_off, (void*)d_keys, (void*)d_vals, (void*)d_keys2, (void*)d_vals2, (void*)d_counts, (void*)d_nruns, d_tmp}) if (q) (void)hipFree(q);
  if (!ok) { for (int* q : {out.pair_ptr, out.pair_i, out.pair_j, out.pair_oa, out.pair_ob}) if (q) (void)hipFree(q); out = PairLists(); }
  return ok;
}

}  // namespace covgpu
