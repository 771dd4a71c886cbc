// What does a chain of small DEPENDENT kernels cost at its boundaries on the MI355X, and what would an in-kernel hand-off buy?
// (A demo-size AIME round is 13 token / pair launches of 10-40 us whose boundaries show 6-7 us gaps in the GPU timeline: 26 boundaries per plan.)
//   mode 0: the chain in ONE stream (what the library does): kernel i + 1 starts when the command processor has retired kernel i
//   mode 1: kernels alternate between TWO streams; kernel i + 1 is launched at once, becomes resident beside kernel i and spins on a device
//           counter that kernel i's workgroups bump when they are done (agent-scope release / acquire around it): the dispatch latency of
//           kernel i + 1 is hidden behind kernel i's run time.  Kernel i + 2 follows kernel i in its stream (normal order) and spins on i + 1.
// Every kernel reads what its predecessor wrote (a 1 MB buffer, checked at the end), busy-works ~`work_us`, writes its own buffer.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/chain_bench.hip -o tools/micro/bin/chain_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void k_link(const float *__restrict__ in, float *__restrict__ out, int n, long long work_cycles,
                                               unsigned *flags, int idx, unsigned wait_count, int handoff, unsigned *err) {
  if (handoff && idx > 0) {
    if (threadIdx.x == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(&flags[idx - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_count) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 24)) { atomicExch(err, 1u); break; }      // never hang the device: report instead
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  extern __shared__ float dyn_lds[];
  if (work_cycles < 0) dyn_lds[threadIdx.x] = 1.f;          // (never: keeps the dynamic LDS allocation alive)
  const long long t0 = wall_clock64();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (int k = i; k < n; k += gridDim.x * blockDim.x) acc += in[k];
  while (wall_clock64() - t0 < work_cycles) acc = acc * 1.0000001f + 1e-9f;
  for (int k = i; k < n; k += gridDim.x * blockDim.x) out[k] = in[k] + 1.0f + (acc == 12345.678f ? 1.f : 0.f);
  if (handoff) {
    __threadfence();                       // agent-scope release of this workgroup's writes
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&flags[idx], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main(int argc, char **argv) {
  const int links = argc > 1 ? atoi(argv[1]) : 26, grid = argc > 2 ? atoi(argv[2]) : 24, n = 1 << 18;
  const double work_us = argc > 3 ? atof(argv[3]) : 15.0;
  const int lds_bytes = argc > 4 ? atoi(argv[4]) : 0, threads = argc > 5 ? atoi(argv[5]) : 256;      // dynamic LDS per workgroup, threads per workgroup
  CK(hipFuncSetAttribute((const void *)k_link, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  int clk_khz = 0;
  CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  const long long work_cycles = (long long)(work_us * 100.0);          // wall_clock64 ticks at 100 MHz
  float *buf[2];
  unsigned *flags, *err;
  CK(hipMalloc(&buf[0], n * sizeof(float))); CK(hipMalloc(&buf[1], n * sizeof(float)));
  CK(hipMalloc(&flags, 256 * sizeof(unsigned))); CK(hipMalloc(&err, sizeof(unsigned)));
  hipStream_t st[2];
  CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  std::vector<float> h(n);
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<double> ms;
    for (int rep = 0; rep < 12; ++rep) {
      CK(hipMemsetAsync(buf[0], 0, n * sizeof(float), st[0]));
      CK(hipMemsetAsync(flags, 0, 256 * sizeof(unsigned), st[0]));
      CK(hipMemsetAsync(err, 0, sizeof(unsigned), st[0]));
      CK(hipStreamSynchronize(st[0]));
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < links; ++i) {
        hipStream_t s = mode ? st[i & 1] : st[0];
        hipLaunchKernelGGL(k_link, dim3(grid), dim3(threads), lds_bytes, s, (const float *)buf[i & 1], buf[(i + 1) & 1], n, work_cycles, flags, i, (unsigned)grid, mode, err);
      }
      CK(hipStreamSynchronize(st[0]));
      if (mode) CK(hipStreamSynchronize(st[1]));
      ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      CK(hipMemcpy(h.data(), buf[links & 1], n * sizeof(float), hipMemcpyDeviceToHost));
      unsigned e = 0;
      CK(hipMemcpy(&e, err, sizeof(e), hipMemcpyDeviceToHost));
      bool ok = !e;
      for (int k = 0; k < n && ok; k += 997) ok = h[k] == (float)links;
      if (!ok) { printf("mode %d rep %d: WRONG RESULT (err %u, h[0] = %g, want %d)\n", mode, rep, e, h[0], links); break; }
    }
    std::sort(ms.begin(), ms.end());
    printf("[lds %d B, %d threads] mode %d (%s): %d links x %d workgroups, ~%.0f us of work each: median %.3f ms  min %.3f ms  -> %.2f us per link beyond the work\n", lds_bytes, threads, mode,
           mode ? "two streams, in-kernel hand-off" : "one stream", links, grid, work_us, ms[ms.size() / 2], ms[0], (ms[0] * 1e3 - links * work_us) / links);
  }
  (void)ev;
  return 0;
}
