// gate_probe.hip — what does a cross-stream dependency cost when it travels through a DEVICE FLAG instead of a HIP event? (dev tool, round 6)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gate_probe.hip -o /tmp/gate_probe && /tmp/gate_probe
// tools/event_probe.hip measured: a record behind a kernel 3.8 us, a cross-stream dependency ~13 us, a record whose waiter is blocked on it holds the
// recording stream's own next kernel ~13 us. Here the same patterns with
//   signal   a one-thread kernel behind the producer that stores a sequence number (agent-scope relaxed store; the producer's data was released by
//            its end-of-kernel fence)
//   gate     a one-wave kernel in front of the consumer that polls the sequence number (s_sleep between polls, bounded) — the consumer kernel
//            behind it starts at a dependent kernel boundary and acquires at its own start
//   tail     the producer kernel itself publishes: every workgroup releases (agent) and draws a ticket, the last one stores the sequence number
//   value    hipStreamWriteValue32 / hipStreamWaitValue32 (command-processor packets, no wave), if the runtime supports them here
// Every mode also CHECKS the payload: the producer writes `seq` over a 2 MB buffer from all XCDs, the consumer counts words that differ.
// Device-side time per link: the whole chain is enqueued behind a 30 ms blocker and timed by events from the blocker's end.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_spin(long long* sink, int cycles) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && sink) *sink = t0;
}
// producer: `nwg` workgroups write seq over the payload, spin to `cycles`; tail != nullptr: publish from inside (ticket of the last workgroup)
__global__ __launch_bounds__(256) void k_produce(int* __restrict__ payload, int nwords, int seq, int cycles, int* ticket, int* flag) {
  const long long t0 = wall_clock64();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nwords; i += gridDim.x * 256) payload[i] = seq;
  while (wall_clock64() - t0 < cycles) {}
  if (flag != nullptr) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == (int)gridDim.x - 1) {
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
__global__ void k_signal(int* flag, int seq) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_gate(const int* flag, int seq, int* timeouts) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) { atomicAdd(timeouts, 1); break; }
    }
  }
}
__global__ __launch_bounds__(256) void k_consume(const int* __restrict__ payload, int nwords, int seq, int* bad, int cycles) {
  const long long t0 = wall_clock64();
  int n = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nwords; i += gridDim.x * 256) n += payload[i] != seq;
  if (n) atomicAdd(bad, n);
  while (wall_clock64() - t0 < cycles) {}
}

int main() {
  const int N = 200, NW = 512 * 1024;   // 2 MB payload
  hipStream_t M, S;
  hipStreamCreateWithFlags(&M, hipStreamNonBlocking); hipStreamCreateWithFlags(&S, hipStreamNonBlocking);
  std::vector<hipEvent_t> e(N), f(N);
  for (auto& x : e) hipEventCreateWithFlags(&x, hipEventDisableTiming);
  for (auto& x : f) hipEventCreateWithFlags(&x, hipEventDisableTiming);
  int *payload, *flags, *bad; long long* sink;
  hipMalloc(&payload, NW * sizeof(int)); hipMalloc(&flags, 4096); hipMalloc(&bad, 64); hipMalloc(&sink, 64);
  hipMemset(flags, 0, 4096); hipMemset(bad, 0, 64);
  int* flagM = flags;        // M -> S sequence
  int* flagS = flags + 64;   // S -> M sequence
  int* ticket = flags + 128;
  int* timeouts = bad + 4;
  int can_value = 0;
  hipDeviceGetAttribute(&can_value, hipDeviceAttributeCanUseStreamWaitValue, 0);
  // signal memory for the value mode
  int* sig = nullptr;
  if (can_value && hipExtMallocWithFlags((void**)&sig, 4096, hipMallocSignalMemory) != hipSuccess) { sig = nullptr; (void)hipGetLastError(); }
  if (sig) hipMemset(sig, 0, 8);
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d, signal memory %s\n", can_value, sig ? "ok" : "unavailable");
  const char* names[] = {"plain (no consumer)", "event: rec + side consumer", "event: round trip M->S->M", "signal + gate: side consumer", "signal + gate: round trip",
                         "tail-publish + gate: side consumer", "tail-publish + gate: round trip", "value: side consumer", "value: round trip"};
  for (int prod_wg : {1, 64}) {
    for (int spin_us : {10}) {
      const int cyc = spin_us * 100;
      for (int mode = 0; mode < 9; ++mode) {
        if (mode >= 7 && !sig) continue;
        double dev = 1e9;
        hipEvent_t t_a, t_b; hipEventCreate(&t_a); hipEventCreate(&t_b);
        int base = 0;
        for (int rep = 0; rep < 3; ++rep) {
          hipDeviceSynchronize();
          hipMemset(flags, 0, 4096); if (sig) hipMemset(sig, 0, 8);
          hipDeviceSynchronize();
          hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, M, sink, 3000000);   // 30 ms
          hipEventRecord(t_a, M);
          for (int i = 0; i < N; ++i) {
            const int seq = base + i + 1;
            const bool round = mode == 2 || mode == 4 || mode == 6 || mode == 8;
            // ---- M waits for the side stream's previous step (round-trip modes)
            if (round && i > 0) {
              if (mode == 2) hipStreamWaitEvent(M, f[i - 1], 0);
              else if (mode == 8) hipStreamWaitValue32(M, sig + 1, (uint32_t)(seq - 1), hipStreamWaitValueGte, 0xffffffffu);
              else hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, M, flagS, seq - 1, timeouts);
            }
            const bool tail = mode == 5 || mode == 6;
            hipLaunchKernelGGL(k_produce, dim3(prod_wg), dim3(256), 0, M, payload, NW, seq, cyc, ticket, tail ? flagM : nullptr);
            if (mode == 1 || mode == 2) hipEventRecord(e[i], M);
            if (mode == 3 || mode == 4) hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, M, flagM, seq);
            if (mode == 7 || mode == 8) hipStreamWriteValue32(M, sig, (uint32_t)seq, 0);
            if (mode == 0) continue;
            // ---- side stream: wait, consume (checks the payload), publish
            if (mode == 1 || mode == 2) hipStreamWaitEvent(S, e[i], 0);
            else if (mode == 7 || mode == 8) hipStreamWaitValue32(S, sig, (uint32_t)seq, hipStreamWaitValueGte, 0xffffffffu);
            else hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, S, flagM, seq, timeouts);
            // (in the round-trip modes the producer of step i+1 waits for this consumer: the payload is stable while it is checked; in the side modes
            //  the producer runs ahead and overwrites it — the check is only meaningful in the round-trip modes)
            hipLaunchKernelGGL(k_consume, dim3(round ? 64 : 1), dim3(256), 0, S, payload, round ? NW : 0, seq, bad, 100);
            if (mode == 2) hipEventRecord(f[i], S);
            if (mode == 4 || mode == 6) hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, S, flagS, seq);
            if (mode == 8) hipStreamWriteValue32(S, sig + 1, (uint32_t)seq, 0);
          }
          hipEventRecord(t_b, M);
          hipStreamSynchronize(M); hipStreamSynchronize(S);
          float ms = 0; hipEventElapsedTime(&ms, t_a, t_b);
          dev = std::min(dev, (double)ms * 1e3 / N);
          base += N;
        }
        int hb[8]; hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost); hipMemset(bad, 0, 64);
        printf("producer %2d WG x %2d us  %-38s device %6.2f us per link (overhead %5.2f)  stale words %d  gate timeouts %d\n", prod_wg, spin_us, names[mode], dev,
               dev - spin_us, hb[0], hb[4]);
      }
    }
  }
  return 0;
}
