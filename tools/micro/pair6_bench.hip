// Stand-alone timing harness of k_pair_t6 (the three-way-split pair kernel, pair_tile6_kernels.hip) on the cfg4 batch shape, beside k_pair_t<1,3>:
// the library kernels and k_pair_t6's timing-only ablations, interleaved rounds in ONE process (median / min per variant).  Synthetic operands, timing only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMIND_PAIR_ABL tools/micro/pair6_bench.hip -o tools/micro/bin/pair6_bench
//   tools/micro/bin/pair6_bench [scenes 24] [N 321] [rounds 7] [update_mode 0]      (PAIR_BENCH_ONLY=substr,substr: a clean A/B of a few variants)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../../mind_amd/csrc/pair_jobs.h"
#include "../../mind_amd/csrc/fusion_kernels.hip"
#include "../../mind_amd/csrc/pair_bf16_kernels.hip"
#include "../../mind_amd/csrc/pair_tile_kernels.hip"
#include "../../mind_amd/csrc/pair_tile6_kernels.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static u32 bf16_bits(float v) { u32 u; memcpy(&u, &v, 4); u += 0x7fffu + ((u >> 16) & 1u); return u >> 16; }
static float bf16_val(u32 b) { u32 u = b << 16; float f; memcpy(&f, &u, 4); return f; }

typedef void (*Kernel3)(const PairJob *, int, float *, const float *, const float *, float *, const u32 *, const u32 *, const float *, const float *,
                        const float *, const float *const *, int);
typedef void (*Kernel6)(const PairJob *, int, float *, const float *, const float *, float *, const u32 *, const u32 *, const u32 *, const u32 *,
                        const float *, const float *, const float *, const float *const *, int);
struct Variant { std::string name; Kernel3 f3; Kernel6 f6; std::vector<float> ms; };

int main(int argc, char **argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 24, N = argc > 2 ? atoi(argv[2]) : 321, rounds = argc > 3 ? atoi(argv[3]) : 7, um = argc > 4 ? atoi(argv[4]) : 0;
  const int a = 64;
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::normal_distribution<float> G(0.f, 1.f);
  const int tiles = (N + 15) / 16, ns = pair_column_splits(N);
  std::vector<PairJob> jobs;
  long long ebt = 0; int ntok = 0, slot = 0;
  for (int b = 0; b < S; ++b) {
    for (int j = 0; j < N; ++j)
      for (int s = 0; s < ns; ++s) {
        PairJob J; memset(&J, 0, sizeof(J));
        J.edge_base_t = ebt; J.N = N; J.j = j; J.t0 = (int)((long long)tiles * s / ns); J.t1 = (int)((long long)tiles * (s + 1) / ns);
        J.tok_base = ntok; J.slot = slot++; J.flags = (j < a || j == N - 1) ? 1 : 0; J.scene = b;
        jobs.push_back(J);
      }
    ntok += N; ebt += (long long)N * tiles * 16;
  }
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int grid = std::min((int)jobs.size(), prop.multiProcessorCount);
  pair_jobs_deal(jobs, grid, PAIR_WAVES, (S >= 8 && grid % 8 == 0) ? 8 : 1);
  const int njobs = (int)jobs.size();
  const size_t edge_floats = (size_t)ebt * 128;
  std::vector<float> h_edge(edge_floats), h_ST((size_t)ntok * 256), h_vt(VT_SIZE), h_rt(1024), h_tp((size_t)ntok * 4);
  for (auto &v : h_edge) v = G(rng);
  for (auto &v : h_ST) v = 0.5f * G(rng);
  for (int i = 0; i < VT_SIZE; ++i) h_vt[i] = (i / 128) % 2 == 0 ? 1.f + 0.1f * U(rng) : 0.1f * U(rng);
  for (auto &v : h_rt) v = 0.3f * U(rng);
  for (auto &v : h_tp) v = U(rng);
  std::vector<u32> h_W(2 * 16384), h_WL(2 * 8192), h_QK((size_t)(ntok + 1) * P6_QK_STRIDE);
  auto split3 = [&](float w, u32 &h, u32 &m, u32 &l) { h = bf16_bits(w); const float r1 = w - bf16_val(h); m = bf16_bits(r1); l = bf16_bits(r1 - bf16_val(m)); };
  for (int mtx = 0; mtx < 2; ++mtx)
    for (int i = 0; i < 8192; ++i) {
      u32 h0, m0, l0, h1, m1, l1;
      split3(0.1f * U(rng), h0, m0, l0); split3(0.1f * U(rng), h1, m1, l1);
      h_W[mtx * 16384 + i] = h0 | (h1 << 16); h_W[mtx * 16384 + 8192 + i] = m0 | (m1 << 16); h_WL[mtx * 8192 + i] = l0 | (l1 << 16);
    }
  for (size_t t = 0; t < (size_t)ntok; ++t)
    for (int i = 0; i < 512; ++i) {
      u32 h0, m0, l0, h1, m1, l1;
      split3(0.3f * G(rng), h0, m0, l0); split3(0.3f * G(rng), h1, m1, l1);
      h_QK[t * P6_QK_STRIDE + i] = h0 | (h1 << 16); h_QK[t * P6_QK_STRIDE + 512 + i] = m0 | (m1 << 16); h_QK[t * P6_QK_STRIDE + 1024 + i] = l0 | (l1 << 16);
    }
  float *d_edge, *d_ST, *d_QK, *d_part, *d_vt, *d_rt, *d_tp; u32 *d_W, *d_WL; PairJob *d_jobs;
  CK(hipMalloc(&d_edge, edge_floats * 4)); CK(hipMalloc(&d_ST, h_ST.size() * 4)); CK(hipMalloc(&d_QK, h_QK.size() * 4));
  CK(hipMalloc(&d_part, (size_t)slot * PART_STRIDE * 4)); CK(hipMalloc(&d_vt, VT_SIZE * 4)); CK(hipMalloc(&d_rt, 4096)); CK(hipMalloc(&d_tp, h_tp.size() * 4));
  CK(hipMalloc(&d_W, h_W.size() * 4)); CK(hipMalloc(&d_WL, h_WL.size() * 4)); CK(hipMalloc(&d_jobs, jobs.size() * sizeof(PairJob)));
  CK(hipMemcpy(d_edge, h_edge.data(), edge_floats * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ST, h_ST.data(), h_ST.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_QK, h_QK.data(), h_QK.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_vt, h_vt.data(), VT_SIZE * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rt, h_rt.data(), 4096, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tp, h_tp.data(), h_tp.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_W, h_W.data(), h_W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_WL, h_WL.data(), h_WL.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_jobs, jobs.data(), jobs.size() * sizeof(PairJob), hipMemcpyHostToDevice));

  std::vector<Variant> vs;
  auto add3 = [&](const char *n, Kernel3 f) { vs.push_back({n, f, nullptr, {}}); };
  auto add6 = [&](const char *n, Kernel6 f) { vs.push_back({n, nullptr, f, {}}); };
  add3("k_pair_t<1,3> (two-way split)", k_pair_t<1, 3, 0>);
  add6("k_pair_t6<1> (three-way split)", k_pair_t6<1, 0>);
#ifdef MIND_PAIR_ABL
  add6("  - lo fragments from LDS instead of L2", k_pair_t6<1, 1>);
  add6("  - no operand splits", k_pair_t6<1, 2>);
  add6("  - no query loads", k_pair_t6<1, 4>);
  add6("  - no T loads", k_pair_t6<1, 8>);
  add6("  - no sum_p_mem", k_pair_t6<1, 16>);
  add6("  - no attention", k_pair_t6<1, 32>);
  add6("  - no gemm2", k_pair_t6<1, 64>);
  add6("  - no gemm1", k_pair_t6<1, 256>);
  add6("  - no gemm1, no gemm2", k_pair_t6<1, 64 + 256>);
  add6("  - no LayerNorms", k_pair_t6<1, 512>);
  add6("  - no edge store", k_pair_t6<1, 1024>);
  add6("  - no edge loads", k_pair_t6<1, 2048>);
  add6("  - GEMMs only (no LN / split / attention / T / q)", k_pair_t6<1, 2 + 4 + 8 + 32 + 512>);
  add6("  - GEMMs only, lo fragments from LDS", k_pair_t6<1, 1 + 2 + 4 + 8 + 32 + 512>);
  add6("  - memory only (no GEMM / LN / split / attention)", k_pair_t6<1, 2 + 32 + 64 + 256 + 512>);
  add6("  - arithmetic only (no loads, no store)", k_pair_t6<1, 4 + 8 + 1024 + 2048>);
  add6("  * half the weight-fragment traffic (LDS and L2)", k_pair_t6<1, 4096>);
  add6("  * half the fragment traffic, no T / q loads", k_pair_t6<1, 4096 + 4 + 8>);
  add6("  * half the fragment traffic, lo from LDS", k_pair_t6<1, 4096 + 1>);
  add6("  - timers", k_pair_t6<1, 128>);
#endif
  add6("k_pair_t6<0> layer 0", k_pair_t6<0, 0>);
  if (const char *only = getenv("PAIR_BENCH_ONLY")) {
    std::vector<std::string> keys;
    for (const char *p = only; *p;) { const char *e = strchr(p, ','); if (!e) e = p + strlen(p); keys.emplace_back(p, e); p = *e ? e + 1 : e; }
    std::vector<Variant> keep;
    for (auto &v : vs)
      for (auto &k : keys) if (v.name.find(k) != std::string::npos) { keep.push_back(v); break; }
    vs.swap(keep);
  }
  const size_t lds = mind_pair_bf_lds_bytes();
  for (auto &v : vs) CK(hipFuncSetAttribute(v.f3 ? (const void *)v.f3 : (const void *)v.f6, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double pairs = (double)S * N * N;
  printf("pair6_bench: %d scenes x N = %d (%d jobs, %d tiles per column in %d splits), %.3f GB of edges each way, grid %d, update_mode %d\n", S, N, njobs, tiles, ns,
         pairs * 512 / 1e9, grid, um);
  for (int r = 0; r < rounds + 1; ++r)
    for (auto &v : vs) {
      if (v.name.find("timers") != std::string::npos && r != 1) continue;
      CK(hipEventRecord(e0, 0));
      if (v.f3)
        hipLaunchKernelGGL(v.f3, dim3(grid), dim3(PAIR_THREADS), lds, 0, d_jobs, njobs, d_edge, d_ST, d_QK, d_part, d_W, d_W + 16384, d_vt, d_rt, d_tp,
                           (const float *const *)nullptr, um);
      else
        hipLaunchKernelGGL(v.f6, dim3(grid), dim3(PAIR_THREADS), lds, 0, d_jobs, njobs, d_edge, d_ST, d_QK, d_part, d_W, d_W + 16384, d_WL, d_WL + 8192, d_vt,
                           d_rt, d_tp, (const float *const *)nullptr, um);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) v.ms.push_back(ms);
    }
  for (auto &v : vs) {
    if (v.ms.empty()) continue;
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
    printf("%-56s median %7.3f ms  min %7.3f ms   %5.2f TB/s edge traffic (%.3f of 8 TB/s)\n", v.name.c_str(), med, mn, pairs * 1024 / (med * 1e-3) / 1e12,
           pairs * 1024 / (med * 1e-3) / 8e12);
  }
  return 0;
}
