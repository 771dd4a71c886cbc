// RelaFusionLayer pair kernel on the bf16 MFMA (v_mfma_f32_16x16x32_bf16), fp32-accurate by operand splitting.
//
// Reference semantics: planners/mind/networks/network.py:165-232 (same math as k_pair in fusion_kernels.hip, whose
// algebraic folds -- rank-decomposed proj_memory, K projection folded into the query, V projection folded out of the
// sum -- are kept).  What changes is the arithmetic of every contraction:
//
//   NP == 3 ("bf16x3"): both operands of a product are split into bf16 hi + lo parts (x = hi + lo + O(2^-17 |x|)) and
//       the three significant partial products hi.hi + hi.lo + lo.hi are accumulated in fp32 on the bf16 MFMA:
//       3 MFMAs of 16 cycles instead of 8 fp32 MFMAs of 32 cycles for the same 16x16x32 block = 5.3 x the fp32-MFMA
//       rate at ~2^-17 relative error per product (measured against the oracle: tests/test_gpu_predictor.py,
//       emulated on the CPU in tests/diag/bf16_split_emulation.py: 7e-6 m on `reg` vs 3e-6 for fp32 itself).
//   NP == 1 ("bf16"): hi parts only (plain bf16 operands, fp32 accumulate): the "stress" arithmetic of BASELINE
//       config 5; misses the 1e-3 m parity bar (3.7e-3 m, same emulation) and is therefore never the default.
//
// Layout algebra (16 pairs per tile, pair p = lane & 15, q = lane >> 4):
//   * a lane holds a pair's 128 features as 8 x f32x4: chunk blk = features 16 blk + 4 q + (0..3) = the MFMA C/D
//     layout (col = lane & 15, row = 4 (lane >> 4) + reg) of output block blk.
//   * B operand of k-group g (32 k-slots; a lane supplies slots 8 q .. 8 q + 7): slot (q, i) <-> feature
//     16 (2 g + (i >> 2)) + 4 q + (i & 3), i.e. chunks 2 g and 2 g + 1 of the lane, packed to bf16 pairs.  The weight
//     fragments are stored in the same slot order (pack_bfrag, mind_hip.hip), so the C/D layout of one GEMM is the B
//     layout of the next: edge -> memory -> proj_edge chain through registers.
//   * LayerNorm means are folded into the weights on the host ((I - 11^T/128) W: LayerNorm is invariant to a shift
//     along the features), so LN1 / LN2 only need the sum of squares.
//   * sum_i p_i mem_i runs on the MFMA too: the memory tile is transposed through LDS as (hi, lo) bf16 pairs, k-slot
//     (q', i) <-> (pair 4 q' + (i >> 1), part i & 1), and contracted with the duplicated probabilities
//     A[head][slot] = p[head][pair]; rows 8..15 of the 16-row A operand are zero (8 heads).
//   * the edge tensor is stored [scene][j][i][128]: a column job streams contiguous memory.
#include <hip/hip_runtime.h>
#include <stdint.h>

// tuning switches (A/B-measured on the MI355X): where the next tile's edge rows and this tile's folded query
// are requested
#ifndef PBF_PREFETCH_TOP
#define PBF_PREFETCH_TOP 0      // 1: next tile's edge rows requested right after this tile's rows left the load registers (measured: no difference)
#endif
// the staging buffers are private to a wave and one wave's DS instructions execute in order: no s_waitcnt between the passes
// over a buffer, only a compiler barrier (draining the LDS queue there measured no difference: DESIGN 4)
#define PBF_FENCE() asm volatile("" ::: "memory")
#ifndef PBF_QK_FIRST
#define PBF_QK_FIRST 1
#endif

#define LB_WE 0                                       // [part 2][ob 8][g 4][lane 64][4 dwords] = 16384 dwords (64 KB)
#define LB_WP 16384
#define LB_STAGE 32768                                // 8 waves x 512 dwords
#define LB_PTAB (LB_STAGE + PAIR_WAVES * STAGE_FLOATS)   // 8 waves x 128: p[head 8][pair 16]
#define LB_SVEC (LB_PTAB + PAIR_WAVES * 128)          // 8 waves x 128: S[j]
#define LB_VT (LB_SVEC + PAIR_WAVES * 128)            // 896
#define LB_RT (LB_VT + VT_SIZE)                       // 1024 (layer 0)
#define LB_TOTAL (LB_RT + 1024)                       // 40832 dwords = 163328 bytes

#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)

// split the lane's 8 chunks into the four k-groups' B operands: hi[g], lo[g] (lo only for NP == 3)
template <int NP>
__device__ __forceinline__ void split_frag(const frag8 &x, u32x4 (&hi)[4], u32x4 (&lo)[4]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const f32x4 v = x[2 * g + c];
      const u32 h0 = pk_bf16(v[0], v[1]), h1 = pk_bf16(v[2], v[3]);
      hi[g][2 * c] = h0;
      hi[g][2 * c + 1] = h1;
      if (NP == 3) {
        lo[g][2 * c] = pk_bf16(v[0] - bf_lo_f32(h0), v[1] - bf_hi_f32(h0));
        lo[g][2 * c + 1] = pk_bf16(v[2] - bf_lo_f32(h1), v[3] - bf_hi_f32(h1));
      }
    }
    SCHED_FENCE();
  }
}

// acc[ob] += sum_g W[ob][g] . B[g]   (W = hi + lo fragments in LDS at `wa`: [part][ob][g][lane][4 dwords]).
// Eight steps (g, output-block quad): per step 4 x NP MFMAs that rotate over four accumulators (dependent MFMAs are
// four issues apart); the hi fragments are double-buffered in registers, the lo fragments are reloaded for the next
// step as soon as the lo.hi products of this one are issued.
template <int NP>
__device__ __forceinline__ void gemm_bf(frag8 &acc, const u32 *wa, const u32x4 (&bhi)[4], const u32x4 (&blo)[4], int lane) {
  int lo_ = lane * 4;
  OPAQUE(lo_);
  const u32 *wl = wa + lo_;
  u32x4 ah[2][4], al[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ah[0][k] = *(const u32x4 *)(wl + ((k * 4 + 0) * 256));
    if (NP == 3) al[k] = *(const u32x4 *)(wl + 8192 + ((k * 4 + 0) * 256));
  }
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int g = st >> 1, o0 = (st & 1) * 4, cur = st & 1, nxt = cur ^ 1;
    const int gn = (st + 1) >> 1, on = ((st + 1) & 1) * 4;
#ifdef MIND_GEMM_NO_RELOAD      // (timing-only build of tools/micro/pair_bench: the weight fragments of the first step serve all eight)
#pragma unroll
    for (int k = 0; k < 4; ++k) { ah[nxt][k] = ah[cur][k]; asm volatile("" : "+v"(ah[nxt][k])); }
#else
    if (st < 7) {
#pragma unroll
      for (int k = 0; k < 4; ++k) ah[nxt][k] = *(const u32x4 *)(wl + (((on + k) * 4 + gn) * 256));
    }
#endif
    if (NP == 3) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[o0 + k] = MFMA_BF(al[k], bhi[g], acc[o0 + k]);
      SCHED_FENCE();
#ifndef MIND_GEMM_NO_RELOAD
      if (st < 7) {
#pragma unroll
        for (int k = 0; k < 4; ++k) al[k] = *(const u32x4 *)(wl + 8192 + (((on + k) * 4 + gn) * 256));
      }
#endif
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[o0 + k] = MFMA_BF(ah[cur][k], bhi[g], acc[o0 + k]);
    if (NP == 3) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[o0 + k] = MFMA_BF(ah[cur][k], blo[g], acc[o0 + k]);
    }
    SCHED_FENCE();
  }
}

// LayerNorm of a shift-free input (the mean was folded into the weights): y = x / sqrt(mean(x^2) + eps) * g + b
__device__ __forceinline__ void ln_nomean(frag8 &a, const float *vtq, int off_g, int off_b) {
  float v = 0.f;
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int w = 0; w < 4; ++w) v = fmaf(a[b][w], a[b][w], v);
  v = red_quad(v);
  const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const f32x4 gm = *(const f32x4 *)(vtq + off_g + 16 * b);
    const f32x4 bt = *(const f32x4 *)(vtq + off_b + 16 * b);
#pragma unroll
    for (int w = 0; w < 4; ++w) a[b][w] = fmaxf(fmaf(a[b][w] * rstd, gm[w], bt[w]), 0.f);
    if (b & 1) SCHED_FENCE();
  }
}

// MODE 0: layer 0, edge built in-kernel from the relative pose encoding; MODE 1: layers 1..5, edge read from HBM.
// update_mode as in k_pair.
template <int MODE, int NP>
__global__ __launch_bounds__(PAIR_THREADS, 1) void k_pair_bf(const PairJob *__restrict__ jobs, int n_jobs, float *__restrict__ edge,
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

  // ---- stage weights / tables (once per workgroup); the lo halves are only needed by the split arithmetic.  All loads of
  //      a matrix are requested before the first LDS write (a load -> wait -> write loop costs 16 L2 round trips per
  //      workgroup, a quarter of a small launch)
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

  float *stage = lds + LB_STAGE + wave * STAGE_FLOATS;
  float *ptab = lds + LB_PTAB + wave * 128;
  float *svec = lds + LB_SVEC + wave * 128;
  const float *vt = lds + LB_VT;
  const u32 *wbe = (const u32 *)(lds + LB_WE);
  const u32 *wbp = (const u32 *)(lds + LB_WP);
  // transposed staging image of a 32-feature quarter of the memory tile: [feature 32][pair 16] dwords (hi | lo << 16),
  // the four pair-quads of a row XOR-swizzled so that both the ds_write_b32 pass and the ds_read_b128 pass are conflict-free
  //   write: lane (p, q) owns features 16 b2 + 4 q + r (quad code of those rows = q)
  //   read : lane (n, q') fetches pairs 4 q' .. 4 q' + 3 of feature 16 b2 + n
  const int cq_w = (0x1320 >> (4 * q)) & 3;                 // c = [0, 2, 3, 1]
  const int cq_r = (0x1320 >> (4 * ((p >> 2) & 3))) & 3;
  const int tw_base = (4 * q) * 16 + (((p >> 2) ^ cq_w) * 4) + (p & 3);
  const int tr_base = p * 16 + ((q ^ cq_r) * 4);
  PT_DECL

  for (int job = wave * gridDim.x + blockIdx.x; job < n_jobs; job += gridDim.x * PAIR_WAVES) {
    const PairJob J = jobs[job];
    if (update_mode == 2 && !(J.flags & 1)) continue;
    if (J.t1 <= J.t0) continue;                  // padding job of the XCD-aware order
    const bool do_update = (update_mode == 0) || (update_mode == 1 && (J.flags & 1));
    const int N = J.N;
    const int j = J.j;
    float *ecol = edge + (((size_t)J.edge_base + (size_t)j * N) << 7);      // this column's pairs are contiguous: [i][128]
    PBF_FENCE();
    if (lane < 32) *(f32x4 *)(svec + lane * 4) = *(const f32x4 *)(ST + (size_t)(J.tok_base + j) * 256 + lane * 4);
    float m_run[4], l_part[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { m_run[k] = -INFINITY; l_part[k] = 0.f; }
    // A operand of the folded query (W_k^T q / 4), written by k_token as bf16 hi / lo fragments: row = head (8 used of
    // 16), k-slot order of the B operand; fetched per tile right before the scores (it would cost 32 registers to keep)
    const u32 *qs = (const u32 *)(QK + (size_t)(J.tok_base + j) * 1024) + ((lane & 7) * 4 + q) * 4;
    const bool qrow = (lane & 15) < 8;
    frag8 mbar;      // [blk][r]: head 4 q + r (q < 2), feature 16 blk + p
#pragma unroll
    for (int b = 0; b < 8; ++b) mbar[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float pj[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
      const f32x4 t = *(const f32x4 *)(tokpos + (size_t)(J.tok_base + j) * 4);
      pj[0] = t[0]; pj[1] = t[1]; pj[2] = t[2]; pj[3] = t[3];
    }
    // first edge tile of the job (later ones are requested one tile ahead)
    f32x4 raw[8];
#define PBF_LOAD_RAW(ROW0, LANEV)                                                                       \
  _Pragma("unroll") for (int qt = 0; qt < 4; ++qt) _Pragma("unroll") for (int n = 0; n < 2; ++n) {     \
    int irow = (ROW0) + 8 * n + ((LANEV) >> 3);                                                         \
    irow = irow < N ? irow : N - 1;                                                                     \
    raw[2 * qt + n] = *(const f32x4 *)(ecol + ((size_t)irow << 7) + qt * 32 + ((LANEV)&7) * 4);        \
  }
    if (MODE == 1) { PBF_LOAD_RAW(J.t0 * 16, lane) }
    PBF_FENCE();

    for (int tile = J.t0; tile < J.t1; ++tile) {
      const int i0 = tile * 16;
      int lq = q, lp = p, ll = lane;
      OPAQUE(lq); OPAQUE(lp); OPAQUE(ll);
      const float *vtq = vt + lq * 4;
      const int i = i0 + p;
      const bool valid = i < N;
      const int ic = valid ? i : (N - 1);
      // accumulator of the first GEMM starts from T[i] (+ S[j] below): requested now, consumed after the edge staging
      frag8 mem;
      {
        const float *Ti = ST + (size_t)(J.tok_base + ic) * 256 + 128 + lq * 4;
#pragma unroll
        for (int b = 0; b < 8; ++b) mem[b] = *(const f32x4 *)(Ti + 16 * b);
      }
      PT(7);                 // (trace) tile top: address setup + issue of the T loads
      PT_DRAIN();
      PT(8);                 // (trace) wait for every outstanding global access: T rows, prefetched edge rows, earlier stores
      frag8 ef;
      if (MODE == 1) {
        // ---- edge tile -> registers through the swizzled staging buffer (full 128-byte row segments)
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
          for (int n = 0; n < 2; ++n) *(f32x4 *)(stage + sw_pos(8 * n + (ll >> 3), ll & 7)) = raw[2 * qt + n];
          PBF_FENCE();
#pragma unroll
          for (int b2 = 0; b2 < 2; ++b2) ef[2 * qt + b2] = *(const f32x4 *)(stage + sw_pos(lp, 4 * b2 + lq));
          PBF_FENCE();
        }
        if (PBF_PREFETCH_TOP && tile + 1 < J.t1) { PBF_LOAD_RAW(i0 + 16, ll) }
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
      PT(0);
      u32x4 mhi[4], mlo[4];
      {
        u32x4 ehi[4], elo[4];
        split_frag<NP>(ef, ehi, elo);
        const float *svq = svec + lq * 4;
#pragma unroll
        for (int b = 0; b < 8; ++b) mem[b] += *(const f32x4 *)(svq + 16 * b);
        SCHED_FENCE();
        gemm_bf<NP>(mem, wbe, ehi, elo, lane);
      }
      PT(1);
      ln_nomean(mem, vtq, VT_GM, VT_BM);
      SCHED_FENCE();
      split_frag<NP>(mem, mhi, mlo);
      PT(2);

      u32x4 qhi[4], qlo[4];
#define PBF_LOAD_QK()                                                                              \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                  \
    qhi[g] = *(const u32x4 *)(qs + g * 128);                                                       \
    if (NP == 3) qlo[g] = *(const u32x4 *)(qs + (4 + g) * 128);                                    \
  }
      // ---- edge update e' = LN_e(e + ReLU(LN(W_p mem + b_p)))   (network.py:201-202)
      if (do_update) {
        frag8 up;
#pragma unroll
        for (int b = 0; b < 8; ++b) up[b] = *(const f32x4 *)(vtq + VT_BP + 16 * b);
        gemm_bf<NP>(up, wbp, mhi, mlo, lane);
        PT(3);
        ln_nomean(up, vtq, VT_GP, VT_BEP);
#pragma unroll
        for (int b = 0; b < 8; ++b) up[b] += ef[b];
        ln_pairs(up, vtq, VT_GE, VT_BE, false);
        SCHED_FENCE();
        // store through the staging buffer: full 128-byte row segments
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
          for (int b2 = 0; b2 < 2; ++b2) *(f32x4 *)(stage + sw_pos(lp, 4 * b2 + lq)) = up[2 * qt + b2];
          PBF_FENCE();
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const int r = 8 * n + (ll >> 3);
            const int cp = ll & 7;
            const f32x4 v = *(const f32x4 *)(stage + sw_pos(r, cp));
            if (i0 + r < N) *(f32x4 *)(ecol + ((size_t)(i0 + r) << 7) + qt * 32 + cp * 4) = v;
          }
          PBF_FENCE();
        }
      }
      SCHED_FENCE();
      PT(4);
      // ---- the folded query first, THEN the next tile's edge rows: the memory counter retires in order, so the wait for the query
      //      fragments below leaves the eight edge-row loads issued behind them in flight (vmcnt(8)) through the whole attention
      //      phase; requested the other way round, that wait drained the prefetch the moment it was issued
#if PBF_QK_FIRST
      PBF_LOAD_QK()
      asm volatile("" ::: "memory");
#endif
#if PBF_QK_FIRST
      // (unconditional: behind a branch the compiler must assume the loads were NOT issued and waits for everything; the job's last tile
      // asks for its own rows again -- L2 hits that nobody reads)
      if (MODE == 1 && !PBF_PREFETCH_TOP) { const int nrow0 = tile + 1 < J.t1 ? i0 + 16 : i0; PBF_LOAD_RAW(nrow0, ll) }
#else
      if (MODE == 1 && !PBF_PREFETCH_TOP && tile + 1 < J.t1) { PBF_LOAD_RAW(i0 + 16, ll) }
#endif
      // ---- attention scores: S^T[head, pair] = QK_j[head, :] . mem^T[:, pair]; the memory tile is already the B operand
      f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f}, sb = sa, sc = sa;
      {
        PT(9);               // (trace) next-tile prefetch issue
#if !PBF_QK_FIRST
        PBF_LOAD_QK()
#endif
        PT_DRAIN();
        PT(10);              // (trace) wait for the folded-query fragments (and the edge stores / prefetch queued before them)
        if (!qrow) {
#pragma unroll
          for (int g = 0; g < 4; ++g) { qhi[g] = (u32x4){0u, 0u, 0u, 0u}; qlo[g] = (u32x4){0u, 0u, 0u, 0u}; }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          sa = MFMA_BF(qhi[g], mhi[g], sa);
          if (NP == 3) {
            sb = MFMA_BF(qhi[g], mlo[g], sb);
            sc = MFMA_BF(qlo[g], mhi[g], sc);
          }
        }
      }
      if (NP == 3) sa += sb + sc;       // lane (p, q): rows 4q..4q+3 = heads 4q..4q+3 of pair p (q < 2)
      // ---- online softmax over i: lanes of quarter q own heads 4q..4q+3
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
      // rescale the running sum p.mem of this lane's heads (rows 4q..4q+3 of the accumulator blocks): skipped when no
      // running maximum of the wave moved (wave-uniform branch)
      if (__any(grew)) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) mbar[b][r] *= scl[r];
      }
      PBF_FENCE();
      // A operand: p[head = lane & 15][pairs 4q .. 4q+3], each duplicated over the (hi, lo) parts of the memory rows
      PT(5);
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
      //      staging buffer one 32-feature quarter (= k-group g) at a time as (hi | lo << 16) dwords
      const u32 *stu = (const u32 *)stage;
      u32 *stw = (u32 *)stage;
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
      PT(6);
    }  // tiles

    // ---- column partial: m[8], l[8], mbar[8][128]
    float *po = part + (size_t)J.slot * PART_STRIDE;
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
  }
#ifdef MIND_PAIR_TRACE
  if (blockIdx.x == 0 && tid == 0)
    printf("[k_pair_bf<%d,%d> um=%d] cycles: tile-top issue %lld | WAIT global (T, edge prefetch, stores) %lld | stage+split %lld | gemm1 %lld | LN1+split %lld | gemm2 %lld | "
           "LN2/3+store %lld | prefetch issue %lld | WAIT query frags (+stores) %lld | scores+softmax %lld | sum_p_mem %lld\n",
           MODE, NP, update_mode, pt_[7], pt_[8], pt_[0], pt_[1], pt_[2], pt_[3], pt_[4], pt_[9], pt_[10], pt_[5], pt_[6]);
#endif
}

#define PAIR_BF_INST(M, NPV)                                                                                           \
  template __global__ void k_pair_bf<M, NPV>(const PairJob *, int, float *, const float *, const float *, float *,     \
                                             const u32 *, const u32 *, const float *, const float *, const float *,    \
                                             const float *const *, int);
PAIR_BF_INST(0, 3)
PAIR_BF_INST(1, 3)
PAIR_BF_INST(0, 1)
PAIR_BF_INST(1, 1)

extern "C" size_t mind_pair_bf_lds_bytes() { return (size_t)LB_TOTAL * sizeof(float); }
