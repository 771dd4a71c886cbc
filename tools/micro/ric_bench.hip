// Micro-benchmark (diagnostic, not product): cycles per node of k_ilqr's two serial loops (Riccati segment sweep, state-chain
// rollout) in isolation, one wave, synthetic 64-node chain.  hipcc --offload-arch=gfx950 -O3 tools/micro/ric_bench.hip
#include "../../mind_amd/csrc/ilqr_kernels.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>
#define NN 64
__global__ __launch_bounds__(IL_THREADS) void k_bench(const IlqrTreeDev *tp, const IlqrConst *cp, long long *cyc, int reps, int active_waves, int pack) {
  extern __shared__ double il_dsm[];
  const IlqrTreeDev T = *tp;
  const IlqrConst &C = *cp;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *scr = il_dsm + (size_t)wave * IL_SCR;
  if (lane < 12) scr[IL_CST + lane] = lane == 0 ? 1.0 : (lane == 6 ? C.dt : 0.0);
  __syncthreads();
  if (wave >= active_waves) return;
  long long t_ric = 0, t_roll = 0;
  for (int rep = 0; rep < reps; ++rep) {
    if (lane < 36) scr[36 + lane] = (lane % 7 == 0) ? 1.0 : 0.01;
    else if (lane < 42) scr[176 + lane - 36] = 0.1;
    IL_WFENCE();
    long long t0 = clock64();
    int sing = il_backward_segment<false>(C, T, T, 0, NN, NN - 1, NN - 2, 1.0, scr);
    long long t1 = clock64();
    il_rollout_packed(C, T, 0, 1, pack, 0, 0);      // `pack` (slot, piece) items in the wave's lanes: the same chain under `pack` mu slots
    long long t2 = clock64();
    t_ric += t1 - t0; t_roll += t2 - t1;
    if (sing) break;
  }
  if (lane == 0) { cyc[wave * 2] = t_ric; cyc[wave * 2 + 1] = t_roll; }
}
int main(int argc, char **argv) {
  const int M = NN;
  std::vector<int> parent(M), seg_start = {0, M}, seg_nodes(M);
  for (int i = 0; i < M; ++i) { parent[i] = i - 1; seg_nodes[i] = i; }
  std::vector<float> prob(M, 1.0f);
  std::vector<double> xs(M * 6), us(M * 2, 0.01), Fx(M * 36, 0.0), Lx(M * 6, 0.1), Lxx(M * 36, 0.0), K(4 * M * 12, 0.01), k(4 * M * 2, 0.01);
  for (int c = 0; c < M; ++c) {
    for (int i = 0; i < 6; ++i) { Fx[c * 36 + i * 7] = 1.0; Lxx[c * 36 + i * 7] = 2.0; xs[c * 6 + i] = 0.1 * i + 0.2 * c; }
    Fx[c * 36 + 2] = 0.19; Fx[c * 36 + 3] = -0.3; Fx[c * 36 + 8] = 0.05; Fx[c * 36 + 9] = 0.9; Fx[c * 36 + 16] = 0.2; Fx[c * 36 + 20] = 0.01; Fx[c * 36 + 23] = 0.4;
    xs[c * 6 + 3] = 0.3; xs[c * 6 + 5] = 0.05; xs[c * 6 + 2] = 5.0;
  }
  auto up = [&](const void *h, size_t n) { void *d; hipMalloc(&d, n); hipMemcpy(d, h, n, hipMemcpyHostToDevice); return d; };
  IlqrTreeDev T; memset(&T, 0, sizeof(T));
  T.M = M; T.n_agents = 1; T.n_levels = M; T.n_segs = 1; T.n_slevels = 1; T.max_level_segs = 1;
  T.parent = (const int *)up(parent.data(), M * 4); T.seg_start = (const int *)up(seg_start.data(), 8); T.seg_nodes = (const int *)up(seg_nodes.data(), M * 4);
  T.prob = (const float *)up(prob.data(), M * 4);
  { const int rec[8] = {0, M, 0, 1, -1, 0, 0, 0}, q1 = M; T.fstep_q0 = (const int *)up(rec, 32); T.fstep_q1 = (const int *)up(&q1, 4); }
  T.xs = (double *)up(xs.data(), M * 48); T.us = (double *)up(us.data(), M * 16); T.Fx = (double *)up(Fx.data(), M * 288); T.Lx = (double *)up(Lx.data(), M * 48);
  T.Lxx = (double *)up(Lxx.data(), M * 288); T.K = (double *)up(K.data(), K.size() * 8); T.k = (double *)up(k.data(), k.size() * 8);
  void *w; hipMalloc(&w, 4 * 10 * M * 64); hipMemset(w, 0, 4 * 10 * M * 64); T.xs_new = (double *)w;
  hipMalloc(&w, 4 * 10 * M * 16); T.us_new = (double *)w;
  IlqrConst C; memset(&C, 0, sizeof(C));
  C.dt = 0.2; C.wb = 2.5; C.w_ctrl[0] = 0.5; C.w_ctrl[1] = 5.0; C.x0[2] = 5.0;
  for (int j = 0; j < IL_NA; ++j) C.alphas[j] = std::pow(1.1, -(double)(j * j));
  IlqrTreeDev *dT = (IlqrTreeDev *)up(&T, sizeof(T)); IlqrConst *dC = (IlqrConst *)up(&C, sizeof(C));
  long long *cyc; hipMalloc(&cyc, 16 * 8);
  const size_t lds = (size_t)IL_WAVES * IL_SCR * 8;
  for (int aw : {1, 4, 5, 8, -1, -4}) {
    const int reps = 20, pack = aw < 0 ? -aw : 1;
    if (aw < 0) aw = 1;
    for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(k_bench, dim3(1), dim3(IL_THREADS), lds, 0, dT, dC, cyc, reps, aw, pack); hipDeviceSynchronize(); }
    long long h[16]; hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
    printf("%d active wave(s), %d item(s) per rollout wave: wave 0 Riccati %.0f cycles / node, rollout %.0f cycles / node", aw, pack, (double)h[0] / reps / NN, (double)h[1] / reps / NN);
    if (aw > 4) printf(" | wave 4 (shares a SIMD with wave 0): %.0f, %.0f", (double)h[8] / reps / NN, (double)h[9] / reps / NN);
    printf("\n");
  }
  return 0;
}
