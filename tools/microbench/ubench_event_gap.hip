// Microbenchmark: what a cross-stream completion event costs on the queue that records it.
// A ~60 us kernel launched back to back on one stream: bare / hipEventRecord behind each / the event attached to the
// kernel's own dispatch (hipExtLaunchKernelGGL stopEvent); and the same with a second stream waiting on every event and
// running a short kernel behind it.   build: hipcc --offload-arch=gfx950 -O2 -o ubench_event_gap ubench_event_gap.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(unsigned long long cycles, unsigned *sink)
{
  const unsigned long long t0 = wall_clock64();
  unsigned v = threadIdx.x;
  while (wall_clock64() - t0 < cycles) v = v * 1664525u + 1013904223u;
  if (v == 0x12345678u) *sink = v;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  unsigned *sink;
  CK(hipMalloc(&sink, 4));
  const int N = 400, R = 32;
  std::vector<hipEvent_t> ev(R);
  for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const unsigned long long long_k = 6000, short_k = 2000; /* wall_clock64 ticks at 100 MHz: 60 us / 20 us */
  auto run = [&](const char *name, int mode, bool second) -> int {
    for (int rep = 0; rep < 3; rep++) {
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < N; k++) {
        hipEvent_t e = ev[k % R];
        if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(125), dim3(64), 0, s1, nullptr, e, 0, long_k, sink);
        else hipLaunchKernelGGL(spin, dim3(125), dim3(64), 0, s1, long_k, sink);
        if (mode == 1) CK(hipEventRecord(e, s1));
        if (second && mode != 0) {
          CK(hipStreamWaitEvent(s2, e, 0));
          hipLaunchKernelGGL(spin, dim3(125), dim3(64), 0, s2, short_k, sink);
        }
      }
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
      if (rep == 2) printf("%-86s %7.2f us per launch\n", name, us);
    }
    return 0;
  };
  if (run("kernel back to back", 0, false)) return 1;
  if (run("hipEventRecord behind each kernel", 1, false)) return 1;
  if (run("event attached to the kernel's dispatch (hipExtLaunchKernelGGL stopEvent)", 2, false)) return 1;
  if (run("hipEventRecord + a second stream waits on it and runs a 20 us kernel", 1, true)) return 1;
  if (run("stopEvent + a second stream waits on it and runs a 20 us kernel", 2, true)) return 1;
  if (run("kernel back to back (again)", 0, false)) return 1;
  return 0;
}
