// Stand-alone timing harness of the pair kernels on the cfg4 batch shape (S scenes x N tokens, full-update layer): the library's
// kernels (k_pair_bf, k_pair_t) and k_pair_t's timing-only ablations, interleaved rounds in ONE process (median / min per variant).
// Synthetic operands (weights U(-0.1, 0.1) split into bf16 hi / lo, edges N(0, 1), T / S / query O(1)): timing only, no parity.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMIND_PAIR_ABL tools/micro/pair_bench.hip -o tools/micro/bin/pair_bench
//   tools/micro/bin/pair_bench [scenes 24] [N 321] [rounds 7] [column splits: default as mind_predict_batch]
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
#ifdef PAIR_BENCH_EXTRA
#include PAIR_BENCH_EXTRA
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static u32 bf16_bits(float v) { u32 u; memcpy(&u, &v, 4); u += 0x7fffu + ((u >> 16) & 1u); return u >> 16; }
static float bf16_val(u32 b) { u32 u = b << 16; float f; memcpy(&f, &u, 4); return f; }

typedef void (*KernelT)(const PairJob *, int, float *, const float *, const float *, float *, const u32 *, const u32 *, const float *,
                        const float *, const float *, const float *const *, int);
struct Variant { std::string name; KernelT fn; bool tiled; int um; std::vector<float> ms; };

int main(int argc, char **argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 24, N = argc > 2 ? atoi(argv[2]) : 321, rounds = argc > 3 ? atoi(argv[3]) : 7;
  const int a = 64;
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::normal_distribution<float> G(0.f, 1.f);
  const int tiles = (N + 15) / 16;
  int ns = pair_column_splits(N);
  if (argc > 4) ns = std::max(1, std::min(atoi(argv[4]), tiles));      // column splits (jobs per column): per-job overhead shows in the difference
  std::vector<PairJob> jobs;
  long long eb = 0, ebt = 0; int ntok = 0, slot = 0;
  for (int b = 0; b < S; ++b) {
    for (int j = 0; j < N; ++j)
      for (int s = 0; s < ns; ++s) {
        PairJob J; memset(&J, 0, sizeof(J));
        J.edge_base = eb; J.edge_base_t = ebt; J.N = N; J.j = j; J.t0 = (int)((long long)tiles * s / ns); J.t1 = (int)((long long)tiles * (s + 1) / ns);
        J.tok_base = ntok; J.slot = slot++; J.flags = (j < a || j == N - 1) ? 1 : 0; J.scene = b;
        jobs.push_back(J);
      }
    ntok += N; eb += (long long)N * N; ebt += (long long)N * tiles * 16;
  }
  {   // the schedule of mind_predict_batch (pair_jobs.h); PAIR_BENCH_PLAIN_ORDER=1: round 3/4's order (scene lanes interleaved, no balancing)
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int g_ = std::min((int)jobs.size(), pr.multiProcessorCount);
    if (getenv("PAIR_BENCH_PLAIN_ORDER") && S >= 8) {
      std::vector<std::vector<PairJob>> lanes(8);
      for (auto &J : jobs) lanes[J.scene % 8].push_back(J);
      size_t mx = 0; for (auto &l : lanes) mx = std::max(mx, l.size());
      PairJob nj; memset(&nj, 0, sizeof(nj)); nj.N = 1;
      std::vector<PairJob> re;
      for (size_t i = 0; i < mx; ++i) for (int x = 0; x < 8; ++x) re.push_back(i < lanes[x].size() ? lanes[x][i] : nj);
      jobs.swap(re);
    } else
      pair_jobs_deal(jobs, g_, PAIR_WAVES, (S >= 8 && g_ % 8 == 0) ? 8 : 1);
  }
  const int njobs = (int)jobs.size();
  const size_t edge_floats = (size_t)ebt * 128;
  std::vector<float> h_edge(edge_floats), h_ST((size_t)ntok * 256), h_vt(VT_SIZE), h_rt(1024), h_tp((size_t)ntok * 4);
  for (auto &v : h_edge) v = G(rng);
  if (const char *fill = getenv("PAIR_BENCH_FILL")) {      // nan | zero: is a launch's time a function of the VALUES it streams?  (timing only)
    const float f = fill[0] == 'n' ? NAN : 0.f;
    for (auto &v : h_edge) v = f;
    printf("edge tensor filled with %s\n", fill);
  }
  for (auto &v : h_ST) v = 0.5f * G(rng);
  for (int i = 0; i < VT_SIZE; ++i) h_vt[i] = (i / 128) % 2 == 0 ? 1.f + 0.1f * U(rng) : 0.1f * U(rng);
  for (auto &v : h_rt) v = 0.3f * U(rng);
  for (auto &v : h_tp) v = U(rng);
  std::vector<u32> h_W(2 * 16384), h_QK((size_t)ntok * 1024);
  for (int m = 0; m < 2; ++m)
    for (int i = 0; i < 8192; ++i) {     // [part][...] dwords: two bf16 each
      const float w0 = 0.1f * U(rng), w1 = 0.1f * U(rng);
      const u32 h0 = bf16_bits(w0), h1 = bf16_bits(w1);
      h_W[m * 16384 + i] = h0 | (h1 << 16);
      h_W[m * 16384 + 8192 + i] = bf16_bits(w0 - bf16_val(h0)) | (bf16_bits(w1 - bf16_val(h1)) << 16);
    }
  for (size_t t = 0; t < (size_t)ntok; ++t)
    for (int i = 0; i < 512; ++i) {
      const float w0 = 0.3f * G(rng), w1 = 0.3f * G(rng);
      const u32 h0 = bf16_bits(w0), h1 = bf16_bits(w1);
      h_QK[t * 1024 + i] = h0 | (h1 << 16);
      h_QK[t * 1024 + 512 + i] = bf16_bits(w0 - bf16_val(h0)) | (bf16_bits(w1 - bf16_val(h1)) << 16);
    }
  float *d_edge, *d_ST, *d_QK, *d_part, *d_vt, *d_rt, *d_tp; u32 *d_W; PairJob *d_jobs;
  CK(hipMalloc(&d_edge, edge_floats * 4)); CK(hipMalloc(&d_ST, h_ST.size() * 4)); CK(hipMalloc(&d_QK, h_QK.size() * 4));
  CK(hipMalloc(&d_part, (size_t)slot * PART_STRIDE * 4)); CK(hipMalloc(&d_vt, VT_SIZE * 4)); CK(hipMalloc(&d_rt, 4096)); CK(hipMalloc(&d_tp, h_tp.size() * 4));
  CK(hipMalloc(&d_W, h_W.size() * 4)); CK(hipMalloc(&d_jobs, jobs.size() * sizeof(PairJob)));
  CK(hipMemcpy(d_edge, h_edge.data(), edge_floats * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ST, h_ST.data(), h_ST.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_QK, h_QK.data(), h_QK.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_vt, h_vt.data(), VT_SIZE * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rt, h_rt.data(), 4096, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tp, h_tp.data(), h_tp.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_W, h_W.data(), h_W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_jobs, jobs.data(), jobs.size() * sizeof(PairJob), hipMemcpyHostToDevice));

  std::vector<Variant> vs;
  auto add = [&](const char *n, KernelT f, bool tiled, int um = 0) { vs.push_back({n, f, tiled, um, {}}); };
  add("k_pair_bf<1,3> (row-major, rounds 2-3)", k_pair_bf<1, 3>, false);
  add("k_pair_t<1,3>", k_pair_t<1, 3, 0>, true);
#ifdef MIND_PAIR_ABL
  add("  - no gemm2", k_pair_t<1, 3, 1>, true);
  add("  - no gemm1", k_pair_t<1, 3, 64>, true);
  add("  - no gemm1, no gemm2", k_pair_t<1, 3, 65>, true);
  add("  - no LayerNorms", k_pair_t<1, 3, 2>, true);
  add("  - no splits", k_pair_t<1, 3, 256>, true);
  add("  - no LN, no splits", k_pair_t<1, 3, 258>, true);
  add("  - no store", k_pair_t<1, 3, 4>, true);
  add("  - no attention", k_pair_t<1, 3, 8>, true);
  add("  - no sum_p_mem", k_pair_t<1, 3, 512>, true);
  add("  - no T loads", k_pair_t<1, 3, 16>, true);
  add("  - no edge loads", k_pair_t<1, 3, 32>, true);
  add("  - no loads, no store", k_pair_t<1, 3, 4 + 16 + 32>, true);
  add("  - GEMMs only (no LN / split / attention)", k_pair_t<1, 3, 2 + 8 + 256>, true);
  add("  - memory only (no GEMM / LN / split / attention)", k_pair_t<1, 3, 1 + 2 + 8 + 64 + 256>, true);
  add("  - timers", k_pair_t<1, 3, 128>, true);
  add("  * half the jobs on waves 0..3 (one wave per SIMD)", k_pair_t<1, 3, 2048>, true);
  add("  * half the jobs on the even waves", k_pair_t<1, 3, 4096>, true);
  add("  * half the jobs, one wave per SIMD, GEMMs only", k_pair_t<1, 3, 2048 + 2 + 8 + 256>, true);
  add("  * half the jobs, one wave per SIMD, memory only", k_pair_t<1, 3, 2048 + 1 + 2 + 8 + 64 + 256>, true);
  add("  - memory only, no T loads", k_pair_t<1, 3, 1 + 2 + 8 + 16 + 64 + 256>, true);
  add("  - memory only, no store", k_pair_t<1, 3, 1 + 2 + 4 + 8 + 64 + 256>, true);
  add("  - column partials stored from the registers (round 4)", k_pair_t<1, 3, 32768>, true);
#endif
  add("k_pair_bf<0,3> layer 0 (row-major)", k_pair_bf<0, 3>, false);
  add("k_pair_t<0,3> layer 0", k_pair_t<0, 3, 0>, true);
#ifdef MIND_PAIR_ABL
  add("  - layer 0 without the edge build (RPE, projection, LayerNorm)", k_pair_t<0, 3, 1024>, true);
  add("  - layer 0, memory only", k_pair_t<0, 3, 1 + 2 + 8 + 64 + 256 + 1024>, true);
#endif
  add("k_pair_bf<1,1> plain bf16 (row-major)", k_pair_bf<1, 1>, false);
  add("k_pair_t<1,1> plain bf16", k_pair_t<1, 1, 0>, true);
#ifdef MIND_PAIR_ABL
#endif
#ifdef PAIR_BENCH_EXTRA_VARIANTS
  PAIR_BENCH_EXTRA_VARIANTS
#endif
  if (const char *only = getenv("PAIR_BENCH_ONLY")) {      // comma-separated substrings: a clean A/B of a few variants (the others leave their edge values behind)
    std::vector<std::string> keys;
    for (const char *a = only; *a;) { const char *e = strchr(a, ','); if (!e) e = a + strlen(a); keys.emplace_back(a, e); a = *e ? e + 1 : e; }
    std::vector<Variant> keep;
    for (auto &v : vs)
      for (auto &k : keys) if (v.name.find(k) != std::string::npos) { keep.push_back(v); break; }
    vs.swap(keep);
  }
  const size_t lds = mind_pair_bf_lds_bytes();
  for (auto &v : vs) CK(hipFuncSetAttribute((const void *)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int grid = std::min(njobs, prop.multiProcessorCount);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double pairs = (double)S * N * N;
  printf("pair_bench: %d scenes x N = %d (%d jobs, %d tiles per column in %d splits), %.3f GB of edges each way, grid %d\n", S, N, njobs, tiles, ns,
         pairs * 512 / 1e9, grid);
  for (int r = 0; r < rounds + 1; ++r)
    for (auto &v : vs) {
      if (v.name.find("timers") != std::string::npos && r != 1) continue;
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(v.fn, dim3(grid), dim3(PAIR_THREADS), lds, 0, d_jobs, njobs, d_edge, d_ST, d_QK, d_part, d_W, d_W + 16384, d_vt, d_rt, d_tp,
                         (const float *const *)nullptr, v.um);
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
#ifdef MIND_PAIR_ABL
  // ---- round-5 variants against their round-4 forms, bit for bit: edge tensor and column partials of one launch from the same operands
  {
    struct Chk { const char *name; KernelT base, alt; int um; };
    const Chk chk[] = {{"k_pair_t<1,3> vs partials stored from the registers", k_pair_t<1, 3, 32768>, k_pair_t<1, 3, 0>, 0},
                       {"k_pair_t<1,1> vs partials stored from the registers", k_pair_t<1, 1, 32768>, k_pair_t<1, 1, 0>, 0}};
    const size_t part_floats = (size_t)slot * PART_STRIDE;
    std::vector<float> e0(edge_floats), e1(edge_floats), p0(part_floats), p1(part_floats);
    for (const Chk &c : chk) {
      for (int v = 0; v < 2; ++v) {
        CK(hipMemcpy(d_edge, h_edge.data(), edge_floats * 4, hipMemcpyHostToDevice));
        CK(hipMemset(d_part, 0, part_floats * 4));
        hipLaunchKernelGGL(v ? c.alt : c.base, dim3(grid), dim3(PAIR_THREADS), lds, 0, d_jobs, njobs, d_edge, d_ST, d_QK, d_part, d_W, d_W + 16384, d_vt, d_rt, d_tp,
                           (const float *const *)nullptr, c.um);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy((v ? e1 : e0).data(), d_edge, edge_floats * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy((v ? p1 : p0).data(), d_part, part_floats * 4, hipMemcpyDeviceToHost));
      }
      size_t de = 0, dp = 0, changed = 0;
      for (size_t i = 0; i < edge_floats; ++i) { de += memcmp(&e0[i], &e1[i], 4) != 0; changed += memcmp(&e0[i], &h_edge[i], 4) != 0; }
      for (size_t i = 0; i < part_floats; ++i) dp += memcmp(&p0[i], &p1[i], 4) != 0;
      printf("bits: %-40s edge words differing %zu of %zu (%zu updated by the launch), partial words differing %zu of %zu\n", c.name, de, edge_floats, changed, dp,
             part_floats);
    }
  }
#endif
  return 0;
}
