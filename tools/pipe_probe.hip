// pipe_probe.hip — what does one hand-over between WORKGROUPS of the same launch cost? (dev tool, round 6; feeds the design of k_bwd_pipe)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pipe_probe.hip -o /tmp/pipe_probe && /tmp/pipe_probe
// A chain of NWG workgroups: workgroup i waits for 128 doubles of workgroup i+1 (the highest index starts), adds one, publishes its own.
//   mode 0  flag + data: agent-scope stores of the data, vmcnt(0), agent-scope flag store | poll the flag, then agent-scope loads of the data
//   mode 1  data only:   the data words themselves are polled against a sentinel (one round trip instead of two)
//   mode 2  as 1, but every workgroup polls ALL earlier publishers in turn the way a column-oriented substitution does (x_q for q = last .. i+1):
//           only the last wait is on the critical path, the others are long satisfied
// stride: workgroups of the chain are `stride` apart in blockIdx (stride 8 = same XCD under round-robin dispatch, 1 = neighbours on different XCDs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

static constexpr unsigned long long kSentinel = 0x7ff8dead0000beefull;

__global__ __launch_bounds__(256) void k_chain(double* xpub, long long* flag, long long seq, int nwg, int stride, int mode, long long* t_out, int* xcc_out) {
  if ((int)blockIdx.x % stride != 0) return;
  const int i = nwg - 1 - (int)blockIdx.x / stride;   // low blockIdx = end of the chain that starts (dispatched first)
  if (i < 0) return;
  const int tid = threadIdx.x;
  __shared__ double sx[128];
  double acc = 0.0;
  const long long t0 = wall_clock64();
  if (i < nwg - 1) {
    const int qlo = i + 1, qhi = mode == 2 ? nwg - 1 : i + 1;
    for (int q = qhi; q >= qlo; --q) {
      if (mode == 0) {
        if (tid == 0) {
          int spins = 0;
          while (__hip_atomic_load(&flag[q * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (tid < 128) acc += __hip_atomic_load(&xpub[q * 128 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (tid < 128) {
          unsigned long long* p = reinterpret_cast<unsigned long long*>(&xpub[q * 128 + tid]);
          unsigned long long v; int spins = 0;
          while ((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kSentinel && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
          acc += __longlong_as_double((long long)v);
        }
      }
    }
  }
  if (tid < 128) sx[tid] = acc + 1.0;
  __syncthreads();
  if (tid < 128) __hip_atomic_store(&xpub[i * 128 + tid], sx[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (mode == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&flag[i * 16], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) {
    t_out[i] = wall_clock64();
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc_out[i] = (int)(xcc & 0xf);
  }
}
__global__ void k_fill(unsigned long long* p, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = kSentinel; }

int main() {
  const int NWG = 16;
  double* xpub; long long *flag, *t_d; int* xcc_d;
  hipMalloc(&xpub, NWG * 128 * 8); hipMalloc(&flag, NWG * 16 * 8); hipMalloc(&t_d, NWG * 8); hipMalloc(&xcc_d, NWG * 4);
  hipMemset(flag, 0, NWG * 16 * 8);
  long long seq = 0;
  for (int stride : {1, 8}) for (int mode = 0; mode < 3; ++mode) {
    double best = 1e9; std::vector<long long> t(NWG); std::vector<int> xc(NWG); double last = 0;
    for (int rep = 0; rep < 20; ++rep) {
      ++seq;
      hipLaunchKernelGGL(k_fill, dim3((NWG * 128 + 255) / 256), dim3(256), 0, 0, reinterpret_cast<unsigned long long*>(xpub), NWG * 128);
      hipLaunchKernelGGL(k_chain, dim3(NWG * stride), dim3(256), 0, 0, xpub, flag, seq, NWG, stride, mode, t_d, xcc_d);
      hipDeviceSynchronize();
      hipMemcpy(t.data(), t_d, NWG * 8, hipMemcpyDeviceToHost); hipMemcpy(xc.data(), xcc_d, NWG * 4, hipMemcpyDeviceToHost);
      const double us = (double)(t[0] - t[NWG - 1]) / 100.0 / (NWG - 1);   // wall_clock64: 100 MHz
      best = std::min(best, us);
      double x0; hipMemcpy(&x0, xpub, 8, hipMemcpyDeviceToHost); last = x0;
    }
    printf("stride %d mode %d: %.2f us per hand-over (x[0] = %.0f)  xcc:", stride, mode, best, last);
    for (int i = 0; i < NWG; ++i) printf(" %d", xc[i]);
    printf("\n");
  }
  return 0;
}
