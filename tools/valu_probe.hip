// valu_probe.hip — issue rate / dependent latency of the few instructions the serial chain of k_potrf_panel is made of (dev tool):
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o _probe/valu_probe && _probe/valu_probe
// One wave alone on a CU; every figure in shader cycles per instruction (clock64 around 64 x 16 instructions).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
#define MEASURE(idx, body)                                              \
  {                                                                     \
    __builtin_amdgcn_s_barrier();                                       \
    long long t0 = clock64();                                           \
    for (int it = 0; it < 64; ++it) { body }                            \
    long long t1 = clock64();                                           \
    if (threadIdx.x == 0) out[idx] = (double)(t1 - t0) / (64.0 * 16.0); \
  }

__global__ __launch_bounds__(64) void k(double* out, const double* in) {
  double a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, m = in[64 + threadIdx.x];
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  __shared__ double lds[256];
  lds[threadIdx.x] = a0;
  // 0: independent v_fmac_f64 (8 accumulators round-robin)
  MEASURE(0, asm volatile("v_fmac_f64 %0, %8, %8\n v_fmac_f64 %1, %8, %8\n v_fmac_f64 %2, %8, %8\n v_fmac_f64 %3, %8, %8\n v_fmac_f64 %4, %8, %8\n v_fmac_f64 %5, %8, %8\n v_fmac_f64 %6, %8, %8\n v_fmac_f64 %7, %8, %8\n"
                          "v_fmac_f64 %0, %8, %8\n v_fmac_f64 %1, %8, %8\n v_fmac_f64 %2, %8, %8\n v_fmac_f64 %3, %8, %8\n v_fmac_f64 %4, %8, %8\n v_fmac_f64 %5, %8, %8\n v_fmac_f64 %6, %8, %8\n v_fmac_f64 %7, %8, %8\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
  // 1: dependent v_fmac_f64
  MEASURE(1, asm volatile(REP16("v_fmac_f64 %0, %1, %1\n") : "+v"(a0) : "v"(m));)
  // 2: independent v_fmac_f64_dpp row_newbcast
  MEASURE(2, asm volatile("v_fmac_f64_dpp %0, %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %1, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                          "v_fmac_f64_dpp %4, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %5, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %6, %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                          "v_fmac_f64_dpp %0, %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %1, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                          "v_fmac_f64_dpp %4, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %5, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %6, %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
  // 3: independent v_mov_b64_dpp
  MEASURE(3, asm volatile(REP16("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "=v"(a1) : "v"(a0));)
  // 4: v_mov_b32_dpp row_newbcast
  MEASURE(4, asm volatile(REP16("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "=v"(i1) : "v"(i0));)
  // 5: v_readlane_b32 (independent, to 4 sgprs)
  MEASURE(5, asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9\n v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9\n"
                          "v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9\n v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9\n"
                          : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(i0));)
  // 6: readlane pair -> fma with sgpr operand, dependent chain through the sgpr only (8 independent accumulators)
  MEASURE(6, asm volatile("v_readlane_b32 s40, %8, 3\n v_readlane_b32 s41, %9, 3\n s_nop 0\n v_fmac_f64 %0, s[40:41], %10\n v_readlane_b32 s42, %8, 4\n v_readlane_b32 s43, %9, 4\n s_nop 0\n v_fmac_f64 %1, s[42:43], %10\n"
                          "v_readlane_b32 s40, %8, 3\n v_readlane_b32 s41, %9, 3\n s_nop 0\n v_fmac_f64 %2, s[40:41], %10\n v_readlane_b32 s42, %8, 4\n v_readlane_b32 s43, %9, 4\n s_nop 0\n v_fmac_f64 %3, s[42:43], %10\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(i1), "v"(m) : "s40", "s41", "s42", "s43");)
  // 7: dependent v_rcp_f64
  MEASURE(7, asm volatile(REP16("v_rcp_f64 %0, %0\n") : "+v"(a0));)
  // 8: ds_bpermute_b32 independent
  MEASURE(8, asm volatile(REP16("ds_bpermute_b32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(i1) : "v"(i2), "v"(i0));)
  // 9: ds_bpermute dependent (latency)
  MEASURE(9, asm volatile(REP16("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(i1) : "v"(i2));)
  // 10: ds_write_b64 + ds_read_b64 round trip (dependent)
  MEASURE(10, asm volatile(REP16("ds_write_b64 %1, %0\n ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "+v"(a0) : "v"(i3 = (threadIdx.x & 63) * 8));)
  // 11: v_mov_b64_dpp -> v_fmac_f64 dependent pair
  MEASURE(11, asm volatile(REP16("v_mov_b64_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64 %0, %1, %2\n") : "+v"(a0), "+v"(a1) : "v"(m));)
  // 12: dependent v_fmac_f64_dpp (own register: needs 2 wait states -> s_nop 1)
  MEASURE(12, asm volatile(REP16("s_nop 1\n v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "+v"(a0) : "v"(m));)
  // 13: v_mfma_f64_16x16x4 dependent
  {
    typedef double v4 __attribute__((ext_vector_type(4)));
    v4 acc = {a0, a1, a2, a3};
    MEASURE(13, acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0);)
    a0 += acc[0] + acc[1] + acc[2] + acc[3];
  }
  // 14: v_mfma_f64_4x4x4 dependent
  MEASURE(14, a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0); a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0); a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0); a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0);
              a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0); a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0); a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0); a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0);
              a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0); a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0); a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0); a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0);
              a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0); a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0); a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0); a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(m, m, a4, 0, 0, 0);)
  // 16: v_mfma_f64_16x16x4, two independent chains interleaved
  {
    typedef double v4 __attribute__((ext_vector_type(4)));
    v4 c0 = {a0, a1, a2, a3}, c1 = {a4, a5, a6, a7};
    MEASURE(16, for (int q = 0; q < 8; ++q) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, c1, 0, 0, 0); })
    a1 += c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
  }
  // 15: ds_read_b64 dependent-address latency
  MEASURE(15, asm volatile(REP16("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(i3));)
  out[32 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i1 + i2 + i3 + s0 + s1 + s2 + s3 + lds[(threadIdx.x + 1) & 63];
}

// sixteen waves (four per SIMD), each a dependent chain of 1024 v_mfma_f64_16x16x4: cycles per instruction as seen by one SIMD
__global__ __launch_bounds__(1024) void k16(double* out, const double* in, int lds_reads) {
  typedef double v4 __attribute__((ext_vector_type(4)));
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 1024) lds[i] = in[i & 63];
  double m = in[threadIdx.x & 63];
  v4 acc = {m, m, m, m};
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
    double a0 = m, a1 = m, a2 = m, a3 = m;
    if (lds_reads) {  // the operand pattern of a tile update: 4 x 16 bytes per lane
      const double2* p = reinterpret_cast<const double2*>(lds + (((threadIdx.x & 15) * 18 + 4 * ((threadIdx.x >> 4) & 3) + 16 * it) & 4095 & ~1));
      const double2 x = p[0], y = p[1];
      a0 = x.x; a1 = x.y; a2 = y.x; a3 = y.y;
    }
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, m, acc, 0, 0, 0);
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) out[0] = (double)(t1 - t0) / (4.0 * 1024.0);
  out[64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  double *out, *in;
  hipMalloc(&out, 4096); hipMalloc(&in, 4096);
  hipMemset(in, 0, 4096);
  for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, in); hipDeviceSynchronize(); }
  double h[32]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[17] = {"v_fmac_f64 independent", "v_fmac_f64 dependent", "v_fmac_f64_dpp independent", "v_mov_b64_dpp independent", "v_mov_b32_dpp independent", "v_readlane_b32 independent",
                        "2 readlane + fmac(sgpr) [per group of 3, /16 counts 4 groups as 16]", "v_rcp_f64 dependent", "ds_bpermute_b32 independent", "ds_bpermute_b32 dependent", "ds_write_b64+ds_read_b64 round trip",
                        "v_mov_b64_dpp -> v_fmac_f64 pair", "s_nop 1 + v_fmac_f64_dpp dependent", "v_mfma_f64_16x16x4 dependent", "v_mfma_f64_4x4x4 dependent", "ds_read_b32 dependent address", "v_mfma_f64_16x16x4 two chains interleaved"};
  for (int i = 0; i < 17; ++i) printf("%-70s %8.1f cycles\n", nm[i], h[i]);
  for (int lr = 0; lr < 2; ++lr) {
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k16, dim3(1), dim3(1024), 0, 0, out, in, lr); hipDeviceSynchronize(); }
    hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    printf("16 waves x 1024 dependent v_mfma_f64_16x16x4%s: %.1f cycles per instruction and SIMD\n", lr ? " + LDS operand reads" : "", h[0]);
  }
  return 0;
}
