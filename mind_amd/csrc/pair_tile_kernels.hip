// RelaFusionLayer pair kernel, tile-native edge tensor: k_pair_t<MODE, NP>.
//
// Reference semantics: planners/mind/networks/network.py:165-232 (RelaFusionLayer), :259-268 (six layers, the last one
// without an edge update).  Arithmetic, algebraic folds and the chained register layouts are those of k_pair_bf
// (pair_bf16_kernels.hip: bf16 hi + lo split operands on v_mfma_f32_16x16x32_bf16, fp32 accumulate).  What changes is how a
// wave gets at its data:
//
//   * the edge tensor lives in HBM in the kernel's own register layout.  A 16-pair tile of column j is one contiguous 8 KB
//     chunk [chunk b 8][lane 64][4 floats]: element (row 16 t + p, feature 16 b + 4 q + r) at ((b 64 + 16 q + p) 4 + r), so
//     the eight loads (and stores) of a tile are eight fully coalesced 1 KB accesses straight into / out of the MFMA C/D
//     layout -- no LDS staging, no transposition passes, no row clamps.  A column holds ceil(N / 16) whole tiles (the rows
//     past N of the last tile are written by layer 0 like any other row, so they are always finite, and masked out of the
//     softmax).  Nobody but these kernels reads the tensor (mind_debug_read un-permutes it for the diagnostics).
//   * the next tile's edge chunk AND its T rows are requested one tile ahead, right behind the edge store, across job
//     boundaries (the next job's descriptor is fetched when a job starts): a tile top no longer waits for anything that was
//     not in flight for a whole attention phase.
//   * the folded query (W_k^T q / 4, bf16 hi / lo fragments written by k_token) stays in 16 registers for the whole
//     column: rows 0..7 of the 16-row A operand carry the hi parts of the 8 heads, rows 8..15 the lo parts, so the scores
//     take 2 MFMAs per k-group instead of 3 and no per-tile reload (the row halves meet through one cross-lane exchange;
//     the three partial products are summed in k_pair_bf's order: the two kernels give the same bits).
//   * the last layer's launch walks a job list that only holds the consumed columns (actors + cls) instead of skipping
//     four jobs out of five after a dependent load each.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_TILE_FLOATS 2048                           // 16 pairs x 128 features
// Plain bf16 arithmetic (NP == 1, BASELINE config 5) keeps the edge tensor in bf16 as well: a tile chunk is the first GEMM's B operand
// as is -- [k-group g 4][lane 64][4 dwords], dword 2 c + h of group g = features 16 (2 g + c) + 4 q + 2 h, + 1 as a bf16 pair -- 4 KB per
// tile, four loads and four stores, no operand conversion on the way in (the mode's operands are rounded to bf16 anyway; half the HBM bytes)
#define PT_EDGE_BF16(NP) ((NP) == 1)

// ABL: timing-only ablation switches of tools/micro/pair_bench.hip (results are wrong with any bit set; the library instantiates 0)
//   1 no second GEMM | 2 no LayerNorms | 4 no edge store | 8 no attention phase | 16 no T loads | 32 no edge loads | 64 no first GEMM
//   128 phase timers (printed by workgroup 0) | 256 no operand splits | 512 no sum_p_mem pass | 1024 layer 0 without its edge build
//   2048 only waves 0..3 work (one per SIMD, half the jobs) | 4096 only the even waves work
//   32768 column partials stored straight from the accumulator registers (round 4)
// Measured with the same harness and dropped (profiles/r04d_pair_bench_experiments.txt; 24 x N = 321, base 0.723 ms): s_setprio 1 for the
// younger wave of every SIMD 0.748, around the GEMMs 0.731, around the VALU phases 0.731; waves 4..7 starting late 0.823; the next tile
// requested behind the second GEMM, before the edge store 0.970 (64 more live registers through two LayerNorms); non-temporal edge loads
// 0.746, stores 0.753; cross-lane reductions on DPP / v_permlane*_swap instead of ds_bpermute 0.735.  Round-4 occupancy probes (same harness,
// profiles/r04t_pair_bench_occupancy.txt): half the jobs on waves 0..3 alone (one wave per SIMD) take 0.541 ms against 0.812 for all jobs on
// eight waves -- a wave runs 1.5 x faster without its SIMD partner, the second wave adds 33 % of throughput; starting waves 4..7 6 k cycles
// late or passing a GEMM token between the two waves of a SIMD (so that their MFMA phases never coincide) gains 4 % (0.762 vs 0.79).
#ifdef MIND_PAIR_ABL
#define PT_ABL(bit) ((ABL & (bit)) != 0)
#define PT_TIME(i) do { if (PT_ABL(128)) { const long long n_ = clock64(); pt_[i] += n_ - pt_t; pt_t = n_; } } while (0)
#else
#define PT_ABL(bit) false
#define PT_TIME(i)
#endif

// Round 5, intra-wave overlap of VALU work with the matrix pipe, measured and dropped.  Program order pinned with
// sched_group_barrier to `MFMA, up to four VALU / LDS, MFMA, ...`: (a) the operand split of the first GEMM's k-groups 1..3 in the shadow of the steps
// before theirs, (b) the whole attention branch (scores, online softmax, sum_p_mem) in the shadow of the second GEMM and of LN_e (248 registers).
// A/B in one process with nothing else launched in between (PAIR_BENCH_ONLY; 24 x N = 321, 3 jobs per column): base 0.709 ms, (a) 0.711, (b) 0.75.
// With two waves per SIMD the partner wave already fills what one wave leaves idle; the launch is bound by its memory-only form (0.53-0.60 ms)
// and its arithmetic-only form (0.67 ms) at the same time.  (profiles/r05y_pair_bench_shadow_experiments.txt shows (a) at 0.707 against 0.741: an
// artefact of the variant's position in the list -- the same kernel times +- 3-5 % by what ran before it (behind the light ablation variants it is
// faster; the VALUES do not matter: NaN / zero / normal edge tensors time the same, profiles/r05z_pair_bench_fill.txt), which is why
// that file's variants are only comparable with their own row of another run.)  What did pay is the schedule of the column jobs: pair_jobs.h.
template <int MODE, int NP, int ABL = 0>
__global__ __launch_bounds__(PAIR_THREADS, 1) void k_pair_t(const PairJob *__restrict__ jobs, int n_jobs, float *__restrict__ edge,
                                                            const float *__restrict__ ST, const float *__restrict__ QK,
                                                            float *__restrict__ part, const u32 *__restrict__ WBe,
                                                            const u32 *__restrict__ WBp, const float *__restrict__ vtab,
                                                            const float *__restrict__ rtab, const float *__restrict__ tokpos,
                                                            const float *const *__restrict__ rpe_ptrs, int update_mode) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int p = lane & 15;
  const int q = lane >> 4;

  // ---- stage weights / tables (once per workgroup): all loads of a matrix in flight before the first LDS write
  {
    constexpr int PER = (NP == 3 ? 4096 : 2048) / PAIR_THREADS;
    f32x4 te[PER], tp[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) te[k] = ((const f32x4 *)WBe)[tid + k * PAIR_THREADS];
    if (update_mode != 2) {
#pragma unroll
      for (int k = 0; k < PER; ++k) tp[k] = ((const f32x4 *)WBp)[tid + k * PAIR_THREADS];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) ((f32x4 *)(lds + LB_WE))[tid + k * PAIR_THREADS] = te[k];
    if (update_mode != 2) {
#pragma unroll
      for (int k = 0; k < PER; ++k) ((f32x4 *)(lds + LB_WP))[tid + k * PAIR_THREADS] = tp[k];
    }
  }
  for (int i = tid; i < VT_SIZE; i += PAIR_THREADS) lds[LB_VT + i] = vtab[i];
  if (MODE == 0)
    for (int i = tid; i < 1024; i += PAIR_THREADS) lds[LB_RT + i] = rtab[i];
  __syncthreads();

  float *stage = lds + LB_STAGE + wave * STAGE_FLOATS;      // transposition image of the sum_p_mem pass
  float *ptab = lds + LB_PTAB + wave * 128;
  float *svec = lds + LB_SVEC + wave * 128;
  const float *vt = lds + LB_VT;
  const u32 *wbe = (const u32 *)(lds + LB_WE);
  const u32 *wbp = (const u32 *)(lds + LB_WP);
  const int cq_w = (0x1320 >> (4 * q)) & 3;                 // c = [0, 2, 3, 1] (see k_pair_bf)
  const int cq_r = (0x1320 >> (4 * ((p >> 2) & 3))) & 3;
  const int tw_base = (4 * q) * 16 + (((p >> 2) ^ cq_w) * 4) + (p & 3);
  const int tr_base = p * 16 + ((q ^ cq_r) * 4);

#ifdef MIND_PAIR_ABL
  long long pt_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt_t = clock64();
  if (PT_ABL(2048) && __builtin_amdgcn_readfirstlane(wave) >= 4) return;      // (timing: one wave per SIMD, half the jobs)
  if (PT_ABL(4096) && (__builtin_amdgcn_readfirstlane(wave) & 1)) return;     // (timing: waves 0, 2, 4, 6 = two per SIMD on two SIMDs)
#endif
  const int stride = gridDim.x * PAIR_WAVES;
  int job = __builtin_amdgcn_readfirstlane(wave * (int)gridDim.x + (int)blockIdx.x);
  f32x4 raw[8], Tn[8];          // the NEXT tile's edge chunk and T rows (requested one tile ahead)
  if (PT_ABL(16) || PT_ABL(32)) {
#pragma unroll
    for (int b = 0; b < 8; ++b) { raw[b] = (f32x4){0.5f, -0.25f, 0.125f, 1.f}; Tn[b] = (f32x4){0.1f, 0.2f, -0.3f, 0.4f}; }
  }
  constexpr bool EB = PT_EDGE_BF16(NP);
  constexpr int TILE_F = EB ? PT_TILE_FLOATS / 2 : PT_TILE_FLOATS;      // dwords per tile chunk
  constexpr int ESH = EB ? 6 : 7;                                       // dwords per pair, log2
  constexpr int NLD = EB ? 4 : 8;                                       // 1 KB accesses per tile
  bool primed = false;          // raw / Tn / s_nx / qa_nx hold the first tile, the S row and the folded query of `job`
  f32x4 s_nx = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 qa_nx[4];

#define PT_LOAD_E(BASE)                                                                                  \
  if (!PT_ABL(32)) { _Pragma("unroll") for (int b = 0; b < NLD; ++b) raw[b] = *(const f32x4 *)((BASE) + b * 256 + lane4); }
#define PT_LOAD_T(TOKB, NN, I0)                                                                          \
  {                                                                                                      \
    int ic_ = (I0) + lp_;                                                                                \
    ic_ = ic_ < (NN) ? ic_ : (NN)-1;                                                                     \
    const float *Ti_ = ST + (size_t)((TOKB) + ic_) * 256 + 128 + lq4_;                                   \
    if (!PT_ABL(16)) { _Pragma("unroll") for (int b = 0; b < 8; ++b) Tn[b] = *(const f32x4 *)(Ti_ + 16 * b); } \
  }

  PairJob Jc = jobs[job < n_jobs ? job : 0];      // this wave's current job record: read one job ahead (a scalar load and its latency per job otherwise)
  for (; job < n_jobs; job += stride) {
    const PairJob J = Jc;
    // the next job of this wave: its first tile is requested during this job's last one
    const int jn = job + stride;
    const bool in_range = jn < n_jobs;
    const PairJob Jn = jobs[in_range ? jn : job];
    Jc = Jn;
    if (J.t1 <= J.t0) { primed = false; continue; }       // padding job (pair_jobs.h)
    const bool has_next = in_range && Jn.t1 > Jn.t0;
    const bool do_update = (update_mode == 0) || (update_mode == 1 && (J.flags & 1));
    const int N = J.N;
    const int j = J.j;
    const int ntile = (N + 15) >> 4;
    float *ecol = edge + (((size_t)J.edge_base_t + (size_t)j * (ntile * 16)) << ESH);
    const float *ecolN = edge + (((size_t)Jn.edge_base_t + (size_t)Jn.j * (((Jn.N + 15) >> 4) * 16)) << ESH);
    PBF_FENCE();
    // S[j] (into the wave's LDS vector) and the folded query A operand (row p < 8 = hi part of head p, row p >= 8 = lo part of head
    // p - 8; zero rows without the split): requested when the previous job of this wave ended, or here
#define PT_LOAD_SQ(JJ)                                                                                                       \
  {                                                                                                                          \
    s_nx = *(const f32x4 *)(ST + (size_t)((JJ).tok_base + (JJ).j) * 256 + (lane & 31) * 4);                                  \
    const u32 *qs_ = (const u32 *)(QK + (size_t)((JJ).tok_base + (JJ).j) * 1024) + (p >> 3) * 512 + ((p & 7) * 4 + q) * 4;   \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) qa_nx[g] = *(const u32x4 *)(qs_ + g * 128);                                \
  }
    if (!primed) PT_LOAD_SQ(J)
    if (lane < 32) *(f32x4 *)(svec + lane * 4) = s_nx;
    u32x4 qa[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) qa[g] = (NP != 3 && p >= 8) ? (u32x4){0u, 0u, 0u, 0u} : qa_nx[g];
    float m_run[4], l_part[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { m_run[k] = -INFINITY; l_part[k] = 0.f; }
    frag8 mbar;      // [blk][r]: head 4 q + r (q < 2), feature 16 blk + p
#pragma unroll
    for (int b = 0; b < 8; ++b) mbar[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float pj[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
      const f32x4 t = *(const f32x4 *)(tokpos + (size_t)(J.tok_base + j) * 4);
      pj[0] = t[0]; pj[1] = t[1]; pj[2] = t[2]; pj[3] = t[3];
    }
    if (!primed) {
      int lp_ = p, lq4_ = q * 4, lane4 = lane * 4;
      OPAQUE(lp_); OPAQUE(lq4_); OPAQUE(lane4);
      if (MODE == 1) { PT_LOAD_E(ecol + (size_t)J.t0 * TILE_F) }
      PT_LOAD_T(J.tok_base, N, J.t0 * 16)
    }
    PBF_FENCE();

    for (int tile = J.t0; tile < J.t1; ++tile) {
      const int i0 = tile * 16;
      int lq = q, lp = p, ll = lane;
      OPAQUE(lq); OPAQUE(lp); OPAQUE(ll);
      const float *vtq = vt + lq * 4;
      const int i = i0 + p;
      const bool valid = i < N;
      const int ic = valid ? i : (N - 1);
      // accumulator of the first GEMM starts from T[i] + S[j]
      frag8 mem, ef;
      PT_TIME(0);              // (timers) previous tile's attention tail / job setup
#pragma unroll
      for (int b = 0; b < 8; ++b) mem[b] = Tn[b];
      if (MODE == 1) {
        if (EB) {
          // bf16 pairs -> fp32 (the residual e + ... of the edge update runs in fp32)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const u32 d0 = __builtin_bit_cast(u32x4, raw[g])[2 * c], d1 = __builtin_bit_cast(u32x4, raw[g])[2 * c + 1];
              ef[2 * g + c] = (f32x4){bf_lo_f32(d0), bf_hi_f32(d0), bf_lo_f32(d1), bf_hi_f32(d1)};
            }
        } else {
#pragma unroll
          for (int b = 0; b < 8; ++b) ef[b] = raw[b];
        }
      } else if (PT_ABL(1024)) {
#pragma unroll
        for (int b = 0; b < 8; ++b) ef[b] = (f32x4){0.25f, 0.5f, 0.f, 1.f} * (float)(lp + 1);      // (timing: no edge build)
      } else {
        // ---- layer 0: edge0 = ReLU(LN(W_r rpe + b_r)), zeros on the cls row / column (network.py:326-330)
        float r5[5];
        const bool is_cls = (ic == N - 1) || (j == N - 1);
        if (rpe_ptrs != nullptr) {
          const float *rp = rpe_ptrs[J.scene];
          const int n1 = N - 1;
          const size_t o = is_cls ? 0 : ((size_t)ic * n1 + j);
#pragma unroll
          for (int k = 0; k < 5; ++k) r5[k] = rp[(size_t)k * n1 * n1 + o];
        } else {
          const f32x4 ti = *(const f32x4 *)(tokpos + (size_t)(J.tok_base + ic) * 4);
          const float dx = pj[0] - ti[0], dy = pj[1] - ti[1];
          const float dist = sqrtf(dx * dx + dy * dy);
          const float nj = sqrtf(pj[2] * pj[2] + pj[3] * pj[3]);
          const float ni = sqrtf(ti[2] * ti[2] + ti[3] * ti[3]);
          const float den1 = nj * ni + 1e-10f;
          const float den2 = nj * dist + 1e-10f;
          r5[0] = (pj[2] * ti[2] + pj[3] * ti[3]) / den1;
          r5[1] = (pj[2] * ti[3] - pj[3] * ti[2]) / den1;
          r5[2] = (pj[2] * dx + pj[3] * dy) / den2;
          r5[3] = (pj[2] * dy - pj[3] * dx) / den2;
          r5[4] = dist * 2.0f / 100.0f;
        }
        const float *rt = lds + LB_RT + lq * 32;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const float *c = rt + b * 128;
          f32x4 acc = *(const f32x4 *)(c + 20);  // bias
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            const f32x4 wk = *(const f32x4 *)(c + 4 * k);
#pragma unroll
            for (int w = 0; w < 4; ++w) acc[w] = fmaf(wk[w], r5[k], acc[w]);
          }
          ef[b] = acc;
        }
        {
          float s = 0.f;
#pragma unroll
          for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int w = 0; w < 4; ++w) s += ef[b][w];
          s = red_quad(s);
          const float mean = s * (1.0f / 128.0f);
          float v = 0.f;
#pragma unroll
          for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int w = 0; w < 4; ++w) { const float d = ef[b][w] - mean; v = fmaf(d, d, v); }
          v = red_quad(v);
          const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
          for (int b = 0; b < 8; ++b) {
            const float *c = rt + b * 128;
            const f32x4 gm = *(const f32x4 *)(c + 24);
            const f32x4 bt = *(const f32x4 *)(c + 28);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const float y = fmaxf((ef[b][w] - mean) * rstd * gm[w] + bt[w], 0.f);
              ef[b][w] = is_cls ? 0.f : y;
            }
          }
        }
      }

      // ---- memory = ReLU(LN(W_e e + S[j] + T[i]))   (network.py:197-199, rank-decomposed, mean folded away)
      u32x4 mhi[4], mlo[4];
      {
        u32x4 ehi[4], elo[4];
        if (PT_ABL(256)) {
#pragma unroll
          for (int g = 0; g < 4; ++g) { ehi[g] = __builtin_bit_cast(u32x4, ef[2 * g]); elo[g] = __builtin_bit_cast(u32x4, ef[2 * g + 1]); }
        } else if (EB && MODE == 1) {
#pragma unroll
          for (int g = 0; g < 4; ++g) ehi[g] = __builtin_bit_cast(u32x4, raw[g]);      // the stored chunk is the B operand
        } else
          split_frag<NP>(ef, ehi, elo);
        const float *svq = svec + lq * 4;
#pragma unroll
        for (int b = 0; b < 8; ++b) mem[b] += *(const f32x4 *)(svq + 16 * b);
        SCHED_FENCE();
        PT_TIME(1);            // (timers) tile top: wait for the prefetched chunk / T rows, split, S[j]
        if (!PT_ABL(64)) gemm_bf<NP>(mem, wbe, ehi, elo, lane);
        else {
#pragma unroll
          for (int g = 0; g < 4; ++g) { mem[2 * g] += __builtin_bit_cast(f32x4, ehi[g]); mem[2 * g + 1] += __builtin_bit_cast(f32x4, elo[g]); }
        }
      }
      PT_TIME(2);              // (timers) first GEMM
      if (!PT_ABL(2)) ln_nomean(mem, vtq, VT_GM, VT_BM);
      SCHED_FENCE();
      if (PT_ABL(256)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { mhi[g] = __builtin_bit_cast(u32x4, mem[2 * g]); mlo[g] = __builtin_bit_cast(u32x4, mem[2 * g + 1]); }
      } else
        split_frag<NP>(mem, mhi, mlo);
      PT_TIME(3);              // (timers) LN1 + split
      // ---- request the next tile (unconditionally: behind a branch the compiler must assume the loads were not issued and
      //      drains the memory counter): the next tile of this column, the first tile of this wave's next job, or -- at the
      //      very end -- this tile once more (cache hits nobody reads)
#define PT_PREFETCH()                                                                                                                       \
      {                                                                                                                                     \
        const bool lastt = tile + 1 == J.t1;                                                                                                \
        const bool nx = lastt && has_next;                                                                                                  \
        const float *pe = nx ? ecolN + (size_t)Jn.t0 * TILE_F : ecol + (size_t)(lastt ? tile : tile + 1) * TILE_F;                          \
        const int pb = nx ? Jn.tok_base : J.tok_base;                                                                                       \
        const int pN = nx ? Jn.N : N;                                                                                                       \
        const int pi0 = nx ? Jn.t0 * 16 : (lastt ? i0 : i0 + 16);                                                                           \
        int lp_ = lp, lq4_ = lq * 4, lane4 = ll * 4;                                                                                        \
        if (MODE == 1) { PT_LOAD_E(pe) }                                                                                                    \
        PT_LOAD_T(pb, pN, pi0)                                                                                                              \
      }


      // ---- edge update e' = LN_e(e + ReLU(LN(W_p mem + b_p)))   (network.py:201-202), stored in place: eight 1 KB stores
      float *etile = ecol + (size_t)tile * TILE_F + ll * 4;
      frag8 up;
      if (do_update) {
#pragma unroll
        for (int b = 0; b < 8; ++b) up[b] = *(const f32x4 *)(vtq + VT_BP + 16 * b);
        if (!PT_ABL(1)) gemm_bf<NP>(up, wbp, mhi, mlo, lane);
        else {
#pragma unroll
          for (int g = 0; g < 4; ++g) { up[2 * g] += __builtin_bit_cast(f32x4, mhi[g]); up[2 * g + 1] += __builtin_bit_cast(f32x4, mlo[g]); }
        }
          PT_TIME(4);            // (timers) second GEMM
      }
      if (do_update) {
        if (!PT_ABL(2)) ln_nomean(up, vtq, VT_GP, VT_BEP);
#pragma unroll
        for (int b = 0; b < 8; ++b) up[b] += ef[b];
        if (!PT_ABL(2)) ln_pairs(up, vtq, VT_GE, VT_BE, false);
        SCHED_FENCE();
        if (!PT_ABL(4)) {
          if (EB) {
            u32x4 uh[4], ul[4];
            split_frag<1>(up, uh, ul);
#pragma unroll
            for (int g = 0; g < 4; ++g) *(u32x4 *)(etile + g * 256) = uh[g];
          } else {
#pragma unroll
            for (int b = 0; b < 8; ++b) *(f32x4 *)(etile + b * 256) = up[b];
          }
        } else {
#pragma unroll
          for (int b = 0; b < 8; ++b) asm volatile("" ::"v"(up[b]));
        }
      }
      SCHED_FENCE();
      PT_TIME(5);              // (timers) LN2 / LN3 + store
      PT_PREFETCH()
      PBF_FENCE();
      PT_TIME(6);              // (timers) issue of the next tile's requests
      if (PT_ABL(8)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { asm volatile("" ::"v"(mhi[g])); asm volatile("" ::"v"(mlo[g])); }
        continue;
      }
      // ---- attention scores: S^T[row, pair] = qa[row, :] . mem^T[:, pair]; rows 0..7 = hi query parts, rows 8..15 = lo parts: the first
      //      chain gives q_hi.m_hi (rows 0..7) and q_lo.m_hi (rows 8..15), the second q_hi.m_lo; summed as k_pair_bf sums its three chains,
      //      hi.hi + (hi.lo + lo.hi), so that the two kernels agree bit for bit (lane quarters 2, 3 carry the padding rows: zero scores)
      f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f}, sb = sa;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        sa = MFMA_BF(qa[g], mhi[g], sa);
        if (NP == 3) sb = MFMA_BF(qa[g], mlo[g], sb);
      }
      if (NP == 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float lo_hi = __shfl_xor(sa[r], 32, 64);      // row 4 (q + 2) + r of the first chain: the lo query part of head 4 q + r
          sa[r] = lq < 2 ? sa[r] + (sb[r] + lo_hi) : 0.f;
        }
      }
      // ---- online softmax over i: lanes of quarter q (and q ^ 2) own heads 4 (q & 1) .. + 3
      float pr[4], scl[4];
      bool grew = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sv = valid ? sa[r] : -INFINITY;
        const float mx = red_max16(sv);
        const float m_new = fmaxf(m_run[r], mx);
        grew = grew || (m_new != m_run[r]);
        scl[r] = __expf(m_run[r] - m_new);
        pr[r] = valid ? __expf(sv - m_new) : 0.f;
        l_part[r] = l_part[r] * scl[r] + pr[r];
        m_run[r] = m_new;
      }
      PBF_FENCE();
      if (lq < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ptab[(4 * lq + r) * 16 + lp] = pr[r];
      }
      if (__any(grew)) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) mbar[b][r] *= scl[r];
      }
      PBF_FENCE();
      // A operand: p[head = lane & 15][pairs 4q .. 4q+3], each duplicated over the (hi, lo) parts of the memory rows
      u32x4 pah, pal;
      {
        f32x4 pv = *(const f32x4 *)(ptab + (lp & 7) * 16 + 4 * lq);
        if (lp >= 8) pv = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const u32 h = pk_bf16(pv[d], pv[d]);
          pah[d] = h;
          if (NP == 3) {
            const float rlo = pv[d] - bf_lo_f32(h);
            pal[d] = pk_bf16(rlo, rlo);
          }
        }
      }
      // ---- mbar[head][f] += sum_pairs p[head][pair] * mem[pair][f] on the MFMA, the memory tile transposed through the
      //      wave's LDS image one 32-feature quarter (= k-group g) at a time as (hi | lo << 16) dwords
      const u32 *stu = (const u32 *)stage;
      u32 *stw = (u32 *)stage;
      PT_TIME(7);              // (timers) scores + softmax
      if (PT_ABL(512)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { asm volatile("" ::"v"(mhi[g])); asm volatile("" ::"v"(mlo[g])); }
        asm volatile("" ::"v"(pah)); asm volatile("" ::"v"(pal));
        continue;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const u32 X = mhi[g][2 * b2 + rr];
            const u32 Y = NP == 3 ? mlo[g][2 * b2 + rr] : 0u;
            stw[tw_base + (16 * b2 + 2 * rr) * 16] = (X & 0xffffu) | (Y << 16);
            stw[tw_base + (16 * b2 + 2 * rr + 1) * 16] = (X >> 16) | (Y & 0xffff0000u);
          }
        PBF_FENCE();
        const u32x4 f0 = *(const u32x4 *)(stu + tr_base);
        const u32x4 f1 = *(const u32x4 *)(stu + tr_base + 256);
        PBF_FENCE();
        mbar[2 * g] = MFMA_BF(pah, f0, mbar[2 * g]);
        mbar[2 * g + 1] = MFMA_BF(pah, f1, mbar[2 * g + 1]);
        if (NP == 3) {
          mbar[2 * g] = MFMA_BF(pal, f0, mbar[2 * g]);
          mbar[2 * g + 1] = MFMA_BF(pal, f1, mbar[2 * g + 1]);
        }
      }
      PT_TIME(8);              // (timers) sum_p_mem
    }  // tiles
    primed = has_next;
    // the next job's S row and folded query travel while this job's column partial is written (at the very end: this job's once more)
    if (has_next) PT_LOAD_SQ(Jn) else PT_LOAD_SQ(J)

    // ---- column partial: m[8], l[8], mbar[8][128], through the wave's idle LDS images so that it leaves as nine coalesced 16-byte stores per
    //      lane instead of forty 4-byte ones from half the lanes (in order behind this job's last LDS reads; a wave's LDS accesses keep their order)
    float *po = part + (size_t)J.slot * PART_STRIDE;
    if (PT_ABL(32768)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float l = red_sum16(l_part[r]);
        if (p == 0 && q < 2) { po[4 * q + r] = m_run[r]; po[8 + 4 * q + r] = l; }
      }
      if (q < 2) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) po[16 + (4 * q + r) * 128 + 16 * b + p] = mbar[b][r];
      }
    } else {
      float lsum[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) lsum[r] = red_sum16(l_part[r]);
      if (p == 0 && q < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { ptab[4 * q + r] = m_run[r]; ptab[8 + 4 * q + r] = lsum[r]; }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {      // heads 4 h .. 4 h + 3 fill the 512-float transposition image
        if (q == h) {
#pragma unroll
          for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[r * 128 + 16 * b + p] = mbar[b][r];
        }
        PBF_FENCE();
        const f32x4 v0 = *(const f32x4 *)(stage + lane * 4), v1 = *(const f32x4 *)(stage + 256 + lane * 4);
        *(f32x4 *)(po + 16 + h * 512 + lane * 4) = v0;
        *(f32x4 *)(po + 16 + h * 512 + 256 + lane * 4) = v1;
        PBF_FENCE();
      }
      if (lane < 4) *(f32x4 *)(po + lane * 4) = *(const f32x4 *)(ptab + lane * 4);
      PBF_FENCE();
    }
  }
#undef PT_LOAD_E
#undef PT_LOAD_T
#undef PT_LOAD_SQ
#undef PT_PREFETCH
#ifdef MIND_PAIR_ABL
  if (PT_ABL(128) && blockIdx.x == 0 && (tid & 63) == 0 && wave < 2)
    printf("[k_pair_t<%d,%d> um=%d wave %d] cycles: tail/setup %lld | tile top (wait, split, S) %lld | gemm1 %lld | LN1+split %lld | gemm2 %lld | LN2/3+store %lld | "
           "prefetch issue %lld | scores+softmax %lld | sum_p_mem %lld\n",
           MODE, NP, update_mode, wave, pt_[0], pt_[1], pt_[2], pt_[3], pt_[4], pt_[5], pt_[6], pt_[7], pt_[8]);
#endif
}

#define PAIR_T_INST(M, NPV)                                                                                            \
  template __global__ void k_pair_t<M, NPV>(const PairJob *, int, float *, const float *, const float *, float *,      \
                                            const u32 *, const u32 *, const float *, const float *, const float *,     \
                                            const float *const *, int);
PAIR_T_INST(0, 3)
PAIR_T_INST(1, 3)
PAIR_T_INST(0, 1)
PAIR_T_INST(1, 1)
