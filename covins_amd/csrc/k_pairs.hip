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
