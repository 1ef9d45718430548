// lat_probe.hip — dependent-chain latencies of the operations the panel kernels' serial parts are made of (dev tool, one wave):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lat_probe.hip -o /tmp/lat_probe && /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ long long tick(double& x) {  // s_memtime ordered after the VALU result x (and before later uses)
  long long t; int dummy;
  int lo = (int)__double_as_longlong(x);
  asm volatile("v_readfirstlane_b32 %1, %2\n\ts_add_u32 %1, %1, 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(dummy) : "v"(lo) : "memory", "scc");
  asm volatile("" : "+v"(x));
  return t;
}
__global__ void k_lat(double* out, long long* cyc, double x0) {
  __shared__ double s[64];
  const int lane = threadIdx.x;
  double x = x0 + lane * 1e-9;
  long long t0, t1;
  // 1: dependent v_fma_f64
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 256; ++i) x = fma(x, 1.0000001, 1e-9);
  t1 = tick(x); if (lane == 0) cyc[0] = t1 - t0;
  // 2: dependent v_mul_f64
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 256; ++i) x = x * 1.0000001;
  t1 = tick(x); if (lane == 0) cyc[1] = t1 - t0;
  // 3: dependent v_rcp_f64
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) x = __builtin_amdgcn_rcp(x);
  t1 = tick(x); if (lane == 0) cyc[2] = t1 - t0;
  // 4: dependent v_rsq_f64
  x = fabs(x) + 1.0;
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) x = __builtin_amdgcn_rsq(x);
  t1 = tick(x); if (lane == 0) cyc[3] = t1 - t0;
  // 5: LDS write -> read (other lane) round trip
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) { s[lane] = x; __builtin_amdgcn_wave_barrier(); x = s[(lane + 1) & 63] + 1.0; __builtin_amdgcn_wave_barrier(); }
  t1 = tick(x); if (lane == 0) cyc[4] = t1 - t0;
  // 6: readlane -> VALU use
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)b, 5), hi = __builtin_amdgcn_readlane((int)(b >> 32), 5);
    x = x + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
  }
  t1 = tick(x); if (lane == 0) cyc[5] = t1 - t0;
  // 7: 8 independent fma chains interleaved (throughput)
  double y[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) y[k] = x + k;
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 64; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] = fma(y[k], 1.0000001, 1e-9);
  x = ((y[0] + y[1]) + (y[2] + y[3])) + ((y[4] + y[5]) + (y[6] + y[7]));
  t1 = tick(x); if (lane == 0) cyc[6] = t1 - t0;
#pragma unroll
  for (int k = 0; k < 8; ++k) x += y[k];
  // 8: ds_bpermute double (2 x b32) -> use
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) * 4, (int)b), hi = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) * 4, (int)(b >> 32));
    x = x + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
  }
  t1 = tick(x); if (lane == 0) cyc[7] = t1 - t0;
  // 9: dependent MFMA f64 chain
  typedef double v4 __attribute__((ext_vector_type(4)));
  t0 = tick(x);
  v4 acc = {x, x, x, x};
#pragma unroll
  for (int i = 0; i < 64; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0);
  x = acc[0];
  t1 = tick(x); if (lane == 0) cyc[8] = t1 - t0;
  // 10: independent MFMA f64 (4 accumulators)
  t0 = tick(x);
  v4 a2[4] = {acc, acc, acc, acc};
  a2[0][0] += x;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) a2[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, a2[k], 0, 0, 0);
  x = (a2[0][0] + a2[1][0]) + (a2[2][0] + a2[3][0]);
  t1 = tick(x); if (lane == 0) cyc[9] = t1 - t0;
  // 11: DPP quad broadcast of a double -> use
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x55, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x55, 0xf, 0xf, false);
    x = x + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
  }
  t1 = tick(x); if (lane == 0) cyc[10] = t1 - t0;
  out[lane] = x + acc[0] + a2[0][0] + a2[1][1] + a2[2][2] + a2[3][3];
}
int main() {
  double* d; long long* c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 16 * 8);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, d, c, 1.5);
  hipDeviceSynchronize();
  long long h[16]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[11] = {"v_fma_f64 dependent", "v_mul_f64 dependent", "v_rcp_f64 dependent", "v_rsq_f64 dependent", "LDS write->read (other lane)+add", "readlane x2 -> add", "v_fma_f64 8 independent chains (per op)", "ds_bpermute x2 -> add", "v_mfma_f64_16x16x4 dependent", "v_mfma_f64_16x16x4 4 independent (per op)", "DPP quad bcast x2 -> add"};
  const int cnt[11] = {256, 256, 64, 64, 64, 64, 512, 64, 64, 64, 64};
  for (int k = 0; k < 11; ++k) printf("%-45s %7.1f cycles (s_memtime ticks) per step\n", nm[k], (double)h[k] / cnt[k]);
  return 0;
}
