// profiles/launch_cost.hip — what one dependent launch, one synchronisation and one polled completion flag cost on this box
// (host side). Build: hipcc --offload-arch=gfx950 -O2 -o launch_cost.bin launch_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void flag_kernel(volatile unsigned* flag, unsigned v) { if (threadIdx.x == 0) *flag = v; }
__global__ void spin_kernel(long long cycles)
{
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
}

static double now_us()
{
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* d = nullptr;
  hipMalloc(&d, 64);
  unsigned* flag = nullptr;
  hipHostMalloc(&flag, 64, hipHostMallocMapped | hipHostMallocCoherent);
  *flag = 0;
  for (int i = 0; i < 2000; ++i)
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, d);
  hipStreamSynchronize(s);
  // 1. host cost of N back-to-back launches (async), and the time until the GPU has run them all
  for (int n : { 1, 4, 8, 16, 64 })
  {
    double best_host = 1e9, best_all = 1e9;
    for (int rep = 0; rep < 50; ++rep)
    {
      const double t0 = now_us();
      for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, d);
      const double t1 = now_us();
      hipStreamSynchronize(s);
      const double t2 = now_us();
      best_host = std::min(best_host, t1 - t0);
      best_all = std::min(best_all, t2 - t0);
    }
    printf("launches %3d: host enqueue %.2f us (%.2f per launch), until synchronised %.2f us (%.2f per launch)\n", n, best_host,
           best_host / n, best_all, best_all / n);
  }
  // 2. the same while the GPU is busy with a 200 us kernel in front (the enqueue is hidden, what is left is GPU-side dispatch)
  for (int n : { 1, 4, 8 })
  {
    double best = 1e9;
    for (int rep = 0; rep < 30; ++rep)
    {
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 20000LL);  // 100 MHz wall clock: 200 us
      const double t0 = now_us();
      for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, d);
      hipStreamSynchronize(s);
      best = std::min(best, now_us() - t0);
    }
    printf("behind a 200 us kernel, %d launches: %.2f us until synchronised (i.e. %.2f us beyond the kernel)\n", n, best, best - 200.0);
  }
  // 3. one kernel + hipStreamSynchronize against one kernel + polling a flag it writes into page-locked memory
  {
    double best_sync = 1e9, best_poll = 1e9;
    for (unsigned rep = 1; rep <= 200; ++rep)
    {
      double t0 = now_us();
      hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag, rep);
      hipStreamSynchronize(s);
      best_sync = std::min(best_sync, now_us() - t0);
    }
    for (unsigned rep = 1000; rep <= 1200; ++rep)
    {
      double t0 = now_us();
      hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag, rep);
      while (*(volatile unsigned*)flag != rep) {}
      best_poll = std::min(best_poll, now_us() - t0);
      hipStreamSynchronize(s);
    }
    printf("one kernel: launch + hipStreamSynchronize %.2f us, launch + polled flag in page-locked memory %.2f us\n", best_sync, best_poll);
    double best_query = 1e9, best_event = 1e9, best_two = 1e9;
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (unsigned rep = 2000; rep <= 2200; ++rep)
    {
      double t0 = now_us();
      hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag, rep);
      while (hipStreamQuery(s) == hipErrorNotReady) {}
      best_query = std::min(best_query, now_us() - t0);
    }
    for (unsigned rep = 3000; rep <= 3200; ++rep)
    {
      double t0 = now_us();
      hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, flag, rep);
      hipEventRecord(ev, s);
      while (hipEventQuery(ev) == hipErrorNotReady) {}
      best_event = std::min(best_event, now_us() - t0);
    }
    for (unsigned rep = 4000; rep <= 4200; ++rep)
    {
      // the library's form: the work kernel, then a one-thread kernel that writes the flag
      double t0 = now_us();
      hipLaunchKernelGGL(empty_kernel, dim3(16), dim3(256), 0, s, d);
      hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, s, flag, rep);
      while (*(volatile unsigned*)flag != rep) {}
      best_two = std::min(best_two, now_us() - t0);
      hipStreamSynchronize(s);
    }
    printf("one kernel: launch + hipStreamQuery spin %.2f us, + hipEventRecord / hipEventQuery spin %.2f us; work kernel + flag kernel + poll %.2f us\n",
           best_query, best_event, best_two);
  }
  // 4. hipGraph of 8 empty kernels in a chain: launch cost
  {
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 8; ++i)
      hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, d);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 5; ++i)
      hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    double best_host = 1e9, best_all = 1e9;
    for (int rep = 0; rep < 50; ++rep)
    {
      const double t0 = now_us();
      hipGraphLaunch(ge, s);
      const double t1 = now_us();
      hipStreamSynchronize(s);
      best_host = std::min(best_host, t1 - t0);
      best_all = std::min(best_all, now_us() - t0);
    }
    printf("hipGraph of 8 chained empty kernels: host %.2f us, until synchronised %.2f us\n", best_host, best_all);
  }
  return 0;
}
