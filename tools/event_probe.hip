// event_probe.hip — what does an event packet between two dependent kernels of one stream cost on this stack? (dev tool)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/event_probe.hip -o /tmp/event_probe && /tmp/event_probe
// A chain of N small dependent kernels (each ~`spin` us of one workgroup) on stream M. Two figures per mode: `host` = wall time per link when the
// host enqueues into an idle stream (what a launch-bound loop sees: the larger of the host's enqueue time and the device's), `device` = the
// same chain enqueued completely BEHIND a 30 ms blocker kernel and timed by events from the blocker's end: the device side alone.
//   plain      nothing between the kernels
//   record     hipEventRecord(e[i], M) behind every kernel (nobody waits)
//   rec+side   ... and a side stream waits for every event and runs a small kernel of its own (the rest-row substitution's pattern)
//   wait_done  hipStreamWaitEvent(M, done) in front of every kernel, `done` recorded long ago on another stream (a satisfied wait)
//   rec+wait   record behind and satisfied wait in front of every kernel (what the panel chain's stream carries per kernel)
//   wait_side  every kernel of M waits for a kernel of the side stream that was enqueued one step earlier (a wait that is NOT yet satisfied)
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
int main() {
  const int N = 400;
  hipStream_t M, S;
  hipStreamCreateWithFlags(&M, hipStreamNonBlocking); hipStreamCreateWithFlags(&S, hipStreamNonBlocking);
  std::vector<hipEvent_t> e(N), f(N);
  for (auto& x : e) hipEventCreateWithFlags(&x, hipEventDisableTiming);
  for (auto& x : f) hipEventCreateWithFlags(&x, hipEventDisableTiming);
  hipEvent_t done; hipEventCreateWithFlags(&done, hipEventDisableTiming);
  long long* sink; hipMalloc(&sink, 64);
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, S, sink, 100); hipEventRecord(done, S); hipDeviceSynchronize();
  for (int spin_us : {2, 10}) {
    const int cyc = spin_us * 100;   // wall_clock64: 100 MHz
    for (int mode = 0; mode < 6; ++mode) {
      double best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
          if (mode == 3 || mode == 4) hipStreamWaitEvent(M, done, 0);
          if (mode == 5 && i > 0) hipStreamWaitEvent(M, f[i - 1], 0);
          hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, M, sink, cyc);
          if (mode == 1 || mode == 2 || mode == 4 || mode == 5) hipEventRecord(e[i], M);
          if (mode == 2 || mode == 5) { hipStreamWaitEvent(S, e[i], 0); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, S, sink + 1, 100); hipEventRecord(f[i], S); }
        }
        hipStreamSynchronize(M); hipStreamSynchronize(S);
        const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / N;
        if (us < best) best = us;
      }
      // device side alone: everything enqueued behind a blocker
      double dev = 1e9;
      hipEvent_t t_a, t_b; hipEventCreate(&t_a); hipEventCreate(&t_b);
      for (int rep = 0; rep < 3; ++rep) {
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, M, sink, 3000000);   // 30 ms
        hipEventRecord(t_a, M);
        for (int i = 0; i < N; ++i) {
          if (mode == 3 || mode == 4) hipStreamWaitEvent(M, done, 0);
          if (mode == 5 && i > 0) hipStreamWaitEvent(M, f[i - 1], 0);
          hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, M, sink, cyc);
          if (mode == 1 || mode == 2 || mode == 4 || mode == 5) hipEventRecord(e[i], M);
          if (mode == 2 || mode == 5) { hipStreamWaitEvent(S, e[i], 0); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, S, sink + 1, 100); hipEventRecord(f[i], S); }
        }
        hipEventRecord(t_b, M);
        hipStreamSynchronize(M); hipStreamSynchronize(S);
        float ms = 0; hipEventElapsedTime(&ms, t_a, t_b);
        dev = std::min(dev, (double)ms * 1e3 / N);
      }
      const char* names[6] = {"plain", "record", "rec+side", "wait_done", "rec+wait", "wait_side"};
      printf("kernel %2d us  %-10s host %6.2f us per link | device %6.2f us per link (boundary + packets = %5.2f)\n", spin_us, names[mode], best, dev, dev - spin_us);
    }
  }
  return 0;
}
