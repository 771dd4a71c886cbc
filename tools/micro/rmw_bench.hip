// What does a read-modify-write stream over 8 KB tile chunks reach on the MI355X, by structure?  (The pair kernel's memory side: every
// wave walks 16-pair tiles = 8 KB chunks, reads a chunk, writes it back in place; k_pair_t's memory-only ablation runs at ~4.4 TB/s.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/rmw_bench.hip -o tools/micro/bin/rmw_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// classic grid-stride in-place scale: every thread 16 B per step
__global__ void k_stream(f32x4 *x, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 v = x[i];
    x[i] = v * 1.0001f;
  }
}
// wave-per-chunk walk: wave w of the launch owns chunks w, w + W, ... (8 KB each = 8 x 1 KB accesses), DEPTH chunks requested ahead,
// ORDER 0: store then next loads, 1: next loads then store; OUT: 0 in place, 1 to a second buffer; JOB: chunks per contiguous run
template <int DEPTH, int ORDER, int OUT>
__global__ void k_chunks(float *x, float *y, int n_chunks, int job) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int W = gridDim.x * (blockDim.x >> 6);
  const int w = wave * gridDim.x + blockIdx.x;
  // chunk sequence of this wave: runs of `job` consecutive chunks, runs dealt round-robin over the waves
  const int runs = n_chunks / job;
  auto chunk_of = [&](int k) { const int r = w + (k / job) * W; return r < runs ? r * job + (k % job) : -1; };
  f32x4 buf[DEPTH][8];
  int k = 0;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int c = chunk_of(d);
    if (c >= 0) {
#pragma unroll
      for (int b = 0; b < 8; ++b) buf[d][b] = *(const f32x4 *)(x + (size_t)c * 2048 + b * 256 + lane * 4);
    }
  }
  for (;; k += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int c = chunk_of(k + d);
      if (c < 0) return;
      f32x4 v[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) v[b] = buf[d][b] * 1.0001f;
      const int cn = chunk_of(k + d + DEPTH);
      const int cl = cn >= 0 ? cn : c;
      float *dst = (OUT ? y : x) + (size_t)c * 2048 + lane * 4;
      if (ORDER == 1) {
#pragma unroll
        for (int b = 0; b < 8; ++b) buf[d][b] = *(const f32x4 *)(x + (size_t)cl * 2048 + b * 256 + lane * 4);
      }
#pragma unroll
      for (int b = 0; b < 8; ++b) *(f32x4 *)(dst + b * 256) = v[b];
      if (ORDER == 0) {
#pragma unroll
        for (int b = 0; b < 8; ++b) buf[d][b] = *(const f32x4 *)(x + (size_t)cl * 2048 + b * 256 + lane * 4);
      }
    }
  }
}

int main(int argc, char **argv) {
  const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) << 20 : (size_t)1280 << 20;
  const int n_chunks = (int)(bytes / 8192);
  float *x, *y;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes));
  std::vector<float> h(bytes / 4, 1.0f);
  CK(hipMemcpy(x, h.data(), bytes, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, auto launch) {
    std::vector<float> ms;
    for (int r = 0; r < 8; ++r) {
      CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (r) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-78s median %.3f ms  %.2f TB/s read + write\n", name, ms[ms.size() / 2], 2.0 * bytes / (ms[ms.size() / 2] * 1e-3) / 1e12);
  };
  printf("rmw_bench: %.2f GB buffer, %d chunks of 8 KB\n", bytes / 1e9, n_chunks);
  timeit("grid-stride in place, 256 x 8 blocks of 256 threads", [&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (f32x4 *)x, bytes / 16); });
  timeit("grid-stride in place, 256 x 32 blocks of 256 threads", [&] { hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, (f32x4 *)x, bytes / 16); });
#define RUN(D, O, OUT, WAVES, JOB, LDS, NAME)                                                                              \
  timeit(NAME, [&] { hipLaunchKernelGGL((k_chunks<D, O, OUT>), dim3(256), dim3(64 * WAVES), LDS, 0, x, y, n_chunks, JOB); });
  RUN(1, 0, 0, 8, 5, 0, "chunks: 8 waves/CU, 1 ahead, store then loads, in place, runs of 5")
  RUN(1, 1, 0, 8, 5, 0, "chunks: 8 waves/CU, 1 ahead, loads then store, in place, runs of 5")
  RUN(2, 1, 0, 8, 5, 0, "chunks: 8 waves/CU, 2 ahead, loads then store, in place, runs of 5")
  RUN(4, 1, 0, 8, 5, 0, "chunks: 8 waves/CU, 4 ahead, loads then store, in place, runs of 5")
  RUN(1, 1, 1, 8, 5, 0, "chunks: 8 waves/CU, 1 ahead, loads then store, OUT of place, runs of 5")
  RUN(1, 1, 0, 8, 1, 0, "chunks: 8 waves/CU, 1 ahead, loads then store, in place, runs of 1")
  RUN(1, 1, 0, 8, 64, 0, "chunks: 8 waves/CU, 1 ahead, loads then store, in place, runs of 64")
  RUN(1, 1, 0, 16, 5, 0, "chunks: 16 waves/CU, 1 ahead, loads then store, in place, runs of 5")
  RUN(2, 1, 0, 16, 5, 0, "chunks: 16 waves/CU, 2 ahead, loads then store, in place, runs of 5")
  RUN(1, 1, 0, 4, 5, 0, "chunks: 4 waves/CU, 1 ahead, loads then store, in place, runs of 5")
  RUN(4, 1, 0, 4, 5, 0, "chunks: 4 waves/CU, 4 ahead, loads then store, in place, runs of 5")
  return 0;
}
