// RelaFusionLayer pair kernel in an fp32-class arithmetic on the bf16 matrix pipe: k_pair_t6<MODE> ("bf16x6").
//
// Reference semantics: planners/mind/networks/network.py:165-232 (RelaFusionLayer), :259-268 (six layers, the last one
// without an edge update).  Data movement, folds, job walk and register layouts are k_pair_t's (pair_tile_kernels.hip:
// tile-native edge tensor, one wave = one column job, prefetch one tile ahead across job boundaries).  The arithmetic of
// every contraction is the reference's fp32 class instead of the two-way split's 16 bits:
//
//   * both operands of a product are split THREE ways into bf16 parts, x = hi + mid + lo.  With round-to-nearest at every
//     step the split of an fp32 number is EXACT (8 + 8 + 8 significand bits; the remainder after hi has at most 16 bits and
//     a magnitude of at most half an ulp of hi, the remainder after mid at most 7 bits), so no operand is rounded at all;
//   * of the nine partial products the six of relative weight >= 2^-16 per operand pair -- hi.hi, hi.mid, mid.hi, mid.mid,
//     hi.lo, lo.hi -- are accumulated in fp32 on v_mfma_f32_16x16x32_bf16 (each product of two bf16 numbers is exact in the
//     fp32 accumulate); the three dropped ones (mid.lo, lo.mid, lo.lo) are below 2^-24 of the term, i.e. below the rounding
//     of the fp32 product the reference itself computes.  Six MFMAs of 16 cycles per 16 x 16 x 32 block against eight fp32
//     MFMAs of 32 cycles: 2.7 x the fp32-MFMA rate (k_pair, fusion_kernels.hip) at its accuracy
//     (tests/test_gpu_predictor.py::test_pair_kernel_arithmetics_meet_the_parity_bar pins the error against the oracle to
//     k_pair's).  k_actor_mfma<6> (actor_mfma_kernels.hip) is the same arithmetic for the ActorNet.
//
// What the third part costs, and where it lives:
//   * weights: the hi and mid fragments of the two 128 x 128 matrices fill the LDS as before (128 KB; mid IS the two-way
//     split's lo part, so WBe / WBp are shared with k_pair_t); a third 32 KB part per matrix does not fit beside them
//     (160 KB per CU), and it feeds one product in six -- its fragments are read from global memory (L2: the same 64 KB for
//     every wave of the device) one GEMM step ahead of their MFMAs, straight into the A operand registers;
//   * activations: the edge tile is split per k-group on the fly inside the first GEMM (the fp32 tile stays in registers
//     for the residual anyway), the memory tile once (three B operands that also feed the scores and sum_p_mem);
//   * folded query: three parts per token (k_token, mode bit 32: [part 3][k-group 4][head 8][lane quarter 4][4 dwords]),
//     fetched per tile behind the second GEMM (two A operands: rows 0..7 / 8..15 = hi / mid parts of the 8 heads, and
//     rows 0..7 = lo parts), four score MFMAs per k-group;
//   * sum_p_mem: the probabilities are split three ways as well; the transposed memory tile goes through the wave's LDS
//     image twice per k-group -- (hi | mid) and (lo | hi) pairs -- and three MFMAs per 16-feature block contract
//     p_hi.(m_hi + m_mid), p_mid.(m_hi + m_mid) and p_hi.m_lo + p_lo.m_hi.
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef P6_RESID_RELOAD
// 1: the next tile's chunk is requested behind the FIRST GEMM (a whole attention phase + GEMM ahead instead of LN_e + store) and the residual
// re-reads this tile's chunk from L2.  Measured and left off: 1.285 against 1.244 ms per 24 x N = 321 launch (profiles/r06e_pair6_bench.txt vs
// r06d): the kernel's vector-memory return path (64 B/clk per CU) already carries 88 KB per tile and wave -- 64 KB of them the lo weight
// fragments -- and 8 KB more cost more than the longer prefetch distance buys
#define P6_RESID_RELOAD 0
#endif
// Also measured and dropped (same harness): a second register set for the next tile's chunk, requested behind the first GEMM -- all eight chunks
// spill inside the tile loop (17 scratch stores + 29 loads per tile), the first two / four chunks alone (8 / 16 registers, 2-3 stores + 2-8 loads per
// tile) run at 1.53-1.58 / 1.64-1.66 ms against 1.43 on the same box (profiles/r06l_pair6_bench_early_prefetch.txt).  And the bound on what a
// 32-pair tile in one 512-register wave per SIMD could buy (every weight fragment -- LDS and L2 -- serving two 16-pair tiles): with every other GEMM
// step re-using the previous step's fragments (timing only, ABL 4096) the launch takes 1.195 against 1.321 ms, -9.5 %, before the loss of the second
// wave's overlap (profiles/r06k_pair6_bench_half_fragment_traffic.txt): not built.
#define P6_QK_STRIDE 1536      // dwords of folded query per token: three parts of 512

// one k-group's three B operands from the lane's chunks 2 g and 2 g + 1
__device__ __forceinline__ void split3_group(const f32x4 &v0, const f32x4 &v1, u32x4 &h, u32x4 &m, u32x4 &l) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const f32x4 v = c ? v1 : v0;
    const u32 h0 = pk_bf16(v[0], v[1]), h1 = pk_bf16(v[2], v[3]);
    const float r0 = v[0] - bf_lo_f32(h0), r1 = v[1] - bf_hi_f32(h0), r2 = v[2] - bf_lo_f32(h1), r3 = v[3] - bf_hi_f32(h1);
    const u32 m0 = pk_bf16(r0, r1), m1 = pk_bf16(r2, r3);
    h[2 * c] = h0; h[2 * c + 1] = h1;
    m[2 * c] = m0; m[2 * c + 1] = m1;
    l[2 * c] = pk_bf16(r0 - bf_lo_f32(m0), r1 - bf_hi_f32(m0));
    l[2 * c + 1] = pk_bf16(r2 - bf_lo_f32(m1), r3 - bf_hi_f32(m1));
  }
}

// acc[ob] += sum_g W[ob][g] . B[g], W = hi + mid fragments in LDS at `wa` ([part 2][ob 8][g 4][lane 64][4 dwords]) + lo fragments in global
// memory at `wg` ([ob 8][g 4][lane 64][4 dwords]); B[g] either given (bh / bm / bl) or split on the fly from the fp32 tile `src` (FLY).
// Eight steps (g, output-block quad) of 24 MFMAs that rotate over four accumulators; every fragment register is reloaded for the next
// step right behind its last MFMA of this one: the lo fragments (global, L2) 20 MFMAs ahead of their use, the LDS ones 12-16 ahead.
// ABL: timing-only ablation switches of tools/micro/pair6_bench.hip (results are wrong with any bit set; the library instantiates 0)
//   1 lo weight fragments re-read from LDS (the mid ones) instead of streamed from L2 | 2 no operand splits | 4 no query loads | 8 no T loads
//   16 no sum_p_mem | 32 no attention at all | 64 no second GEMM | 128 phase timers (printed by workgroup 0) | 256 no first GEMM | 512 no LayerNorms
//   1024 no edge store | 2048 no edge loads | 4096 every other GEMM step re-uses the previous step's weight fragments (half the LDS and L2 fragment traffic:
//   what a 32-pair tile would pay per pair)
#ifdef MIND_PAIR_ABL
#define P6_ABL(bit) ((ABL & (bit)) != 0)
#define P6_TIME(i) do { if (P6_ABL(128)) { const long long n_ = clock64(); pt_[i] += n_ - pt_t; pt_t = n_; } } while (0)
#else
#define P6_ABL(bit) false
#define P6_TIME(i)
#endif
#define P6_WL(FRAG) (P6_ABL(1) ? *(const u32x4 *)(wl + 8192 + (FRAG) * 256) : __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wg, vo_, (FRAG) * 1024, 0)))
// Steps of (g, NOB output blocks), 6 NOB MFMAs each, that rotate over NOB accumulators.  The hi / mid fragments (LDS) are reloaded for the next
// step right behind their last MFMA of this one; the lo fragments (global memory: L2) travel RING steps ahead in a register ring.
// P6_GEMM_NOB 2: sixteen steps, lo ring two deep (32 fragment registers); 4: eight steps, lo fragments one step ahead (48 fragment registers).
#ifndef P6_GEMM_NOB
#define P6_GEMM_NOB 2
#endif
template <bool FLY, int ABL>
__device__ __forceinline__ void gemm6(frag8 &acc, const u32 *wa, __amdgpu_buffer_rsrc_t wg, const frag8 &src, const u32x4 (&bh4)[4],
                                      const u32x4 (&bm4)[4], const u32x4 (&bl4)[4], int lane, const float *trow = nullptr, f32x4 *Tn = nullptr) {
#ifndef P6_LO_RING
// steps the lo fragments travel ahead.  Measured 3 / 4 deep (24 x N = 321 launch, profiles/r06aq_pair6_bench_lo_ring.txt): 1.39 / 1.40 against 1.42-1.43 ms,
// but both need 256 VGPRs + 32-40 B of scratch per wave (a private segment on every launch, the 30 us demo-size ones included): left at 2
#define P6_LO_RING (P6_GEMM_NOB == 2 ? 2 : 1)
#endif
  constexpr int NOB = P6_GEMM_NOB, SPG = 8 / NOB, NST = 4 * SPG, RING = P6_LO_RING;
  int lo_ = lane * 4;
  OPAQUE(lo_);
  const u32 *wl = wa + lo_;
  const int vo_ = lo_ * 4;      // byte offset of the lane's 16 bytes inside a 1 KB fragment (buffer load: one VGPR of address for all 32 fragments)
  u32x4 ah[NOB], am[NOB], al[RING][NOB];
  // step st: g = st / SPG, output blocks o0 = NOB (st % SPG) .. + NOB - 1
#define P6_G(ST) ((ST) / SPG)
#define P6_O(ST) (NOB * ((ST) % SPG))
#pragma unroll
  for (int k = 0; k < NOB; ++k) {
#pragma unroll
    for (int r = 0; r < RING; ++r) al[r][k] = P6_WL((P6_O(r) + k) * 4 + P6_G(r));
    am[k] = *(const u32x4 *)(wl + 8192 + ((P6_O(0) + k) * 4 + P6_G(0)) * 256);
    ah[k] = *(const u32x4 *)(wl + ((P6_O(0) + k) * 4 + P6_G(0)) * 256);
  }
  // (first GEMM) the T[i] rows of the tile: requested BEHIND the first steps' lo fragments -- the memory counter retires in order, a wait for
  // those fragments would otherwise wait for the rows (L2 / HBM) as well; consumed behind the GEMM
  if (FLY && trow != nullptr && !P6_ABL(8)) {
#pragma unroll
    for (int b = 0; b < 8; ++b) Tn[b] = *(const f32x4 *)(trow + 16 * b);
  }
  u32x4 bh, bm, bl;
#pragma unroll
  for (int st = 0; st < NST; ++st) {
    const int g = P6_G(st), o0 = P6_O(st), cur = st % RING;
    if (FLY) {
      if (st % SPG == 0) {
        if (P6_ABL(2)) { bh = __builtin_bit_cast(u32x4, src[2 * g]); bm = __builtin_bit_cast(u32x4, src[2 * g + 1]); bl = bh ^ bm; }
        else split3_group(src[2 * g], src[2 * g + 1], bh, bm, bl);
      }
    } else {
      bh = bh4[g]; bm = bm4[g]; bl = bl4[g];
    }
#pragma unroll
    for (int k = 0; k < NOB; ++k) acc[o0 + k] = MFMA_BF(al[cur][k], bh, acc[o0 + k]);
    SCHED_FENCE();
    if (st + RING < NST && !(P6_ABL(4096) && ((st + RING) & 2))) {
#pragma unroll
      for (int k = 0; k < NOB; ++k) al[cur][k] = P6_WL((P6_O(st + RING) + k) * 4 + P6_G(st + RING));
    }
#pragma unroll
    for (int k = 0; k < NOB; ++k) acc[o0 + k] = MFMA_BF(am[k], bm, acc[o0 + k]);
#pragma unroll
    for (int k = 0; k < NOB; ++k) acc[o0 + k] = MFMA_BF(am[k], bh, acc[o0 + k]);
    SCHED_FENCE();
    if (st + 1 < NST && !(P6_ABL(4096) && ((st + 1) & 1))) {
#pragma unroll
      for (int k = 0; k < NOB; ++k) am[k] = *(const u32x4 *)(wl + 8192 + ((P6_O(st + 1) + k) * 4 + P6_G(st + 1)) * 256);
    }
#pragma unroll
    for (int k = 0; k < NOB; ++k) acc[o0 + k] = MFMA_BF(ah[k], bl, acc[o0 + k]);
#pragma unroll
    for (int k = 0; k < NOB; ++k) acc[o0 + k] = MFMA_BF(ah[k], bm, acc[o0 + k]);
#pragma unroll
    for (int k = 0; k < NOB; ++k) acc[o0 + k] = MFMA_BF(ah[k], bh, acc[o0 + k]);
    SCHED_FENCE();
    if (st + 1 < NST && !(P6_ABL(4096) && ((st + 1) & 1))) {
#pragma unroll
      for (int k = 0; k < NOB; ++k) ah[k] = *(const u32x4 *)(wl + ((P6_O(st + 1) + k) * 4 + P6_G(st + 1)) * 256);
    }
  }
#undef P6_G
#undef P6_O
}

template <int MODE, int ABL = 0>
__global__ __launch_bounds__(PAIR_THREADS, 1) void k_pair_t6(const PairJob *__restrict__ jobs, int n_jobs, float *__restrict__ edge,
                                                             const float *__restrict__ ST, const float *__restrict__ QK,
                                                             float *__restrict__ part, const u32 *__restrict__ WBe,
                                                             const u32 *__restrict__ WBp, const u32 *__restrict__ WLe,
                                                             const u32 *__restrict__ WLp, const float *__restrict__ vtab,
                                                             const float *__restrict__ rtab, const float *__restrict__ tokpos,
                                                             const float *const *__restrict__ rpe_ptrs, int update_mode) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int p = lane & 15;
  const int q = lane >> 4;

  // ---- stage the hi + mid weight fragments / tables (once per workgroup): all loads of a matrix in flight before the first LDS write
  {
    constexpr int PER = 4096 / PAIR_THREADS;
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

  // the lo fragments of the two matrices as buffer resources (32 KB each): wave-uniform base + one VGPR of lane offset for every fragment
  const __amdgpu_buffer_rsrc_t rle = __builtin_amdgcn_make_buffer_rsrc((void *)WLe, 0, 32768, 0x00020000);
  const __amdgpu_buffer_rsrc_t rlp = __builtin_amdgcn_make_buffer_rsrc((void *)WLp, 0, 32768, 0x00020000);
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
#endif
  const int stride = gridDim.x * PAIR_WAVES;
  int job = __builtin_amdgcn_readfirstlane(wave * (int)gridDim.x + (int)blockIdx.x);
  f32x4 raw[8];                 // the NEXT tile's edge chunk (requested one tile ahead)
  bool primed = false;          // raw / s_nx hold the first tile and the S row of `job`
  f32x4 s_nx = (f32x4){0.f, 0.f, 0.f, 0.f};

#define P6_LOAD_E(BASE)                                                                                  \
  if (!P6_ABL(2048)) { _Pragma("unroll") for (int b = 0; b < 8; ++b) raw[b] = *(const f32x4 *)((BASE) + b * 256 + lane4); }
// the next tile's edge chunk: the next tile of this column, the first tile of this wave's next job, or -- at the very end -- this tile once more
#define P6_PREFETCH(COND)                                                                                                                   \
      if (MODE == 1 && (COND)) {                                                                                                            \
        const bool lastt = tile + 1 == J.t1;                                                                                                \
        const bool nx = lastt && has_next;                                                                                                  \
        const float *pe = nx ? ecolN + (size_t)Jn.t0 * PT_TILE_FLOATS : ecol + (size_t)(lastt ? tile : tile + 1) * PT_TILE_FLOATS;          \
        int lane4 = ll * 4;                                                                                                                 \
        P6_LOAD_E(pe)                                                                                                                       \
      }
#define P6_LOAD_S(JJ) { s_nx = *(const f32x4 *)(ST + (size_t)((JJ).tok_base + (JJ).j) * 256 + (lane & 31) * 4); }

  PairJob Jc = jobs[job < n_jobs ? job : 0];      // this wave's current job record: read one job ahead
  for (; job < n_jobs; job += stride) {
    const PairJob J = Jc;
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
    float *ecol = edge + (((size_t)J.edge_base_t + (size_t)j * (ntile * 16)) << 7);
    const float *ecolN = edge + (((size_t)Jn.edge_base_t + (size_t)Jn.j * (((Jn.N + 15) >> 4) * 16)) << 7);
    PBF_FENCE();
    if (!primed) P6_LOAD_S(J)
    if (lane < 32) *(f32x4 *)(svec + lane * 4) = s_nx;
    // folded query of this column: A operand rows p < 8 = hi part of head p, p >= 8 = mid part of head p - 8 (qa1); rows p < 8 = lo part (qa2)
    const u32 *qk1 = (const u32 *)QK + (size_t)(J.tok_base + j) * P6_QK_STRIDE + (p >> 3) * 512 + ((p & 7) * 4 + q) * 4;
    // (lanes p >= 8 read 512 dwords past their mid part for qa2 -- the next token's record, or the buffer's slack -- and are zeroed below)
    const u32 *qk2 = qk1 + 1024;
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
      if (MODE == 1) { P6_LOAD_E(ecol + (size_t)J.t0 * PT_TILE_FLOATS) }
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
      // the first GEMM accumulates onto S[j]; the T[i] rows (L2: every column of the scene reads them) are requested inside it and added behind it
      // (layers 1..5: the fp32 edge tile IS the prefetched chunk -- one set of registers, requested again where the tile's last reader is done)
      frag8 mem, ef0;
      f32x4 (&ef)[8] = MODE == 1 ? raw : ef0;
      f32x4 Tn[8];
      P6_TIME(0);              // (timers) previous tile's tail / job set-up
      const float *Ti_ = ST + (size_t)(J.tok_base + ic) * 256 + 128 + lq * 4;
      if (P6_ABL(8) || P6_ABL(256)) {
#pragma unroll
        for (int b = 0; b < 8; ++b) Tn[b] = (f32x4){0.1f, 0.2f, -0.3f, 0.4f};
      }
      if (MODE == 0) {
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
      u32x4 mh[4], mm[4], ml[4];
      {
        const float *svq = svec + lq * 4;
#pragma unroll
        for (int b = 0; b < 8; ++b) mem[b] = *(const f32x4 *)(svq + 16 * b);
        SCHED_FENCE();
        P6_TIME(1);            // (timers) tile top: wait for the prefetched chunk, S[j]
        if (!P6_ABL(256)) gemm6<true, ABL>(mem, wbe, rle, ef, mh, mm, ml, lane, Ti_, Tn);      // (mh / mm / ml are not read by the FLY form)
        else {
#pragma unroll
          for (int b = 0; b < 8; ++b) mem[b] += ef[b];
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) mem[b] += Tn[b];
      }
      SCHED_FENCE();
      P6_TIME(2);              // (timers) first GEMM (+ wait for the T rows)
      // the folded query of this column for this tile's scores (in flight through LN1 and the split): A operand rows p < 8 = hi part of head p,
      // p >= 8 = mid part of head p - 8 (qa1); rows p < 8 = lo part (qa2).  Requested BEFORE the next tile's chunk (in-order memory counter: the
      // scores must not wait for an HBM round trip)
      u32x4 qa1[4], qa2[4];
      if (P6_ABL(4)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { qa1[g] = (u32x4){0x3f803f80u, 0x3e803e80u, 0x3f003f00u, 0x3f803f80u}; qa2[g] = qa1[g]; asm volatile("" : "+v"(qa1[g]), "+v"(qa2[g])); }
      } else {
        int qo = 0;
        OPAQUE(qo);
#pragma unroll
        for (int g = 0; g < 4; ++g) { qa1[g] = *(const u32x4 *)(qk1 + qo + g * 128); qa2[g] = *(const u32x4 *)(qk2 + qo + g * 128); }
      }
      // the next tile's edge chunk, into the registers of this tile's fp32 chunk where the first GEMM was its last reader: a tile without an edge
      // update (or every tile under P6_RESID_RELOAD, see the top of the file); a tile with one keeps the chunk for the residual and requests behind it
      P6_PREFETCH(P6_RESID_RELOAD || !do_update)
      if (!P6_ABL(512)) ln_nomean(mem, vtq, VT_GM, VT_BM);
      SCHED_FENCE();
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (P6_ABL(2)) { mh[g] = __builtin_bit_cast(u32x4, mem[2 * g]); mm[g] = __builtin_bit_cast(u32x4, mem[2 * g + 1]); ml[g] = mh[g] ^ mm[g]; }
        else split3_group(mem[2 * g], mem[2 * g + 1], mh[g], mm[g], ml[g]);
        SCHED_FENCE();
      }
      P6_TIME(3);              // (timers) LN1 + split
      if (P6_ABL(32)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { asm volatile("" ::"v"(qa1[g]), "v"(qa2[g])); }
      } else {
      // ---- attention scores S^T[row, pair] = qa[row, :] . mem^T[:, pair].  qa1 rows 0..7 / 8..15 = hi / mid query parts, qa2 rows 0..7 = lo parts
      //      (rows 8..15 zeroed): sa = (q_hi | q_mid).m_hi, sb = (q_hi | q_mid).m_mid, sc = q_hi.m_lo (its rows 8..15, q_mid.m_lo, are
      //      below 2^-24 and not used), sd = q_lo.m_hi; summed smallest first
      f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f}, sb = sa, sc = sa, sd = sa;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u32x4 a2 = lp < 8 ? qa2[g] : (u32x4){0u, 0u, 0u, 0u};
        sd = MFMA_BF(a2, mh[g], sd);
        sc = MFMA_BF(qa1[g], ml[g], sc);
        sb = MFMA_BF(qa1[g], mm[g], sb);
        sa = MFMA_BF(qa1[g], mh[g], sa);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float mid_hi = __shfl_xor(sa[r], 32, 64);      // row 4 (q + 2) + r: the mid query part of head 4 q + r against m_hi
        const float mid_mid = __shfl_xor(sb[r], 32, 64);     // ... against m_mid
        sa[r] = lq < 2 ? sa[r] + ((sb[r] + mid_hi) + ((sc[r] + sd[r]) + mid_mid)) : 0.f;
      }
      SCHED_FENCE();
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
      // A operands: p[head = lane & 15][pairs 4q .. 4q+3], k-slot (q', i) <-> (pair 4 q' + (i >> 1), part i & 1) of the staged image:
      //   pa_hh = (p_hi, p_hi) and pa_mm = (p_mid, p_mid) against the (m_hi | m_mid) image, pa_hl = (p_hi, p_lo) against the (m_lo | m_hi) image
      u32x4 pa_hh, pa_mm, pa_hl;
      {
        f32x4 pv = *(const f32x4 *)(ptab + (lp & 7) * 16 + 4 * lq);
        if (lp >= 8) pv = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const u32 h = pk_bf16(pv[d], pv[d]);
          const float r1 = pv[d] - bf_lo_f32(h);
          const u32 m = pk_bf16(r1, r1);
          const float r2 = r1 - bf_lo_f32(m);
          pa_hh[d] = h;
          pa_mm[d] = m;
          pa_hl[d] = pk_bf16(bf_lo_f32(h), r2);
        }
      }
      P6_TIME(4);              // (timers) scores + softmax
      // ---- mbar[head][f] += sum_pairs p[head][pair] * mem[pair][f] on the MFMA, the memory tile transposed through the wave's LDS
      //      image one 32-feature quarter (= k-group g) at a time, twice: (hi | mid << 16) and (lo | hi << 16) dwords
      if (P6_ABL(16)) {
        asm volatile("" ::"v"(pa_hh), "v"(pa_mm), "v"(pa_hl));
      } else {
        const u32 *stu = (const u32 *)stage;
        u32 *stw = (u32 *)stage;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const u32 X = mh[g][2 * b2 + rr], Y = mm[g][2 * b2 + rr];
              stw[tw_base + (16 * b2 + 2 * rr) * 16] = (X & 0xffffu) | (Y << 16);
              stw[tw_base + (16 * b2 + 2 * rr + 1) * 16] = (X >> 16) | (Y & 0xffff0000u);
            }
          PBF_FENCE();
          const u32x4 f0 = *(const u32x4 *)(stu + tr_base);
          const u32x4 f1 = *(const u32x4 *)(stu + tr_base + 256);
          PBF_FENCE();
#pragma unroll
          for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const u32 X = ml[g][2 * b2 + rr], Y = mh[g][2 * b2 + rr];
              stw[tw_base + (16 * b2 + 2 * rr) * 16] = (X & 0xffffu) | (Y << 16);
              stw[tw_base + (16 * b2 + 2 * rr + 1) * 16] = (X >> 16) | (Y & 0xffff0000u);
            }
          PBF_FENCE();
          const u32x4 f2 = *(const u32x4 *)(stu + tr_base);
          const u32x4 f3 = *(const u32x4 *)(stu + tr_base + 256);
          PBF_FENCE();
          mbar[2 * g] = MFMA_BF(pa_hl, f2, mbar[2 * g]);
          mbar[2 * g + 1] = MFMA_BF(pa_hl, f3, mbar[2 * g + 1]);
          mbar[2 * g] = MFMA_BF(pa_mm, f0, mbar[2 * g]);
          mbar[2 * g + 1] = MFMA_BF(pa_mm, f1, mbar[2 * g + 1]);
          mbar[2 * g] = MFMA_BF(pa_hh, f0, mbar[2 * g]);
          mbar[2 * g + 1] = MFMA_BF(pa_hh, f1, mbar[2 * g + 1]);
          SCHED_FENCE();
        }
      }
      }      // (attention)
      P6_TIME(5);              // (timers) sum_p_mem

      // ---- edge update e' = LN_e(e + ReLU(LN(W_p mem + b_p)))   (network.py:201-202), stored in place: eight 1 KB stores.  Last in the tile: the
      //      three memory operands die with the second GEMM, so the LayerNorms and the store run beside 100 free registers
      if (do_update) {
        float *etile = ecol + (size_t)tile * PT_TILE_FLOATS + ll * 4;
        frag8 up;
#pragma unroll
        for (int b = 0; b < 8; ++b) up[b] = *(const f32x4 *)(vtq + VT_BP + 16 * b);
        if (!P6_ABL(64)) gemm6<false, ABL>(up, wbp, rlp, up, mh, mm, ml, lane);
        else {
#pragma unroll
          for (int g = 0; g < 4; ++g) { up[2 * g] += __builtin_bit_cast(f32x4, mh[g]); up[2 * g + 1] += __builtin_bit_cast(f32x4, mm[g] ^ ml[g]); }
        }
        SCHED_FENCE();
        P6_TIME(6);            // (timers) second GEMM
        f32x4 e2[8];           // this tile's fp32 chunk once more, for the residual (in flight through LN2)
        if (MODE == 1 && P6_RESID_RELOAD && !P6_ABL(2048)) {
#pragma unroll
          for (int b = 0; b < 8; ++b) e2[b] = *(const f32x4 *)(etile + b * 256);
        } else {
#pragma unroll
          for (int b = 0; b < 8; ++b) e2[b] = (f32x4){0.5f, -0.25f, 0.125f, 1.f};
        }
        if (!P6_ABL(512)) ln_nomean(up, vtq, VT_GP, VT_BEP);
        if (MODE == 1 && P6_RESID_RELOAD) {
#pragma unroll
          for (int b = 0; b < 8; ++b) up[b] += e2[b];
        } else {
#pragma unroll
          for (int b = 0; b < 8; ++b) up[b] += ef[b];
        }
        SCHED_FENCE();
        P6_PREFETCH(!P6_RESID_RELOAD)      // (P6_RESID_RELOAD 0) the fp32 tile is dead: its registers take the next tile's chunk
        if (!P6_ABL(512)) ln_pairs(up, vtq, VT_GE, VT_BE, false);
        SCHED_FENCE();
        if (!P6_ABL(1024)) {
#pragma unroll
          for (int b = 0; b < 8; ++b) *(f32x4 *)(etile + b * 256) = up[b];
        } else {
#pragma unroll
          for (int b = 0; b < 8; ++b) asm volatile("" ::"v"(up[b]));
        }
        P6_TIME(7);            // (timers) LN2 / LN_e + store
      } else if (!P6_ABL(64)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" ::"v"(mh[g]), "v"(mm[g]), "v"(ml[g]));
      }
      SCHED_FENCE();
    }  // tiles
    primed = has_next;
    // the next job's S row travels while this job's column partial is written (at the very end: this job's once more)
    if (has_next) P6_LOAD_S(Jn) else P6_LOAD_S(J)

    // ---- column partial: m[8], l[8], mbar[8][128], through the wave's idle LDS images as nine coalesced 16-byte stores per lane
    float *po = part + (size_t)J.slot * PART_STRIDE;
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
#ifdef MIND_PAIR_ABL
  if (P6_ABL(128) && blockIdx.x == 0 && (tid & 63) == 0 && wave < 2)
    printf("[k_pair_t6<%d> um=%d wave %d] cycles: tail/setup %lld | tile top (wait, S) %lld | gemm1 + T wait %lld | LN1+split %lld | scores+softmax %lld | sum_p_mem %lld | "
           "gemm2 %lld | LN2/LN_e+store %lld\n", MODE, update_mode, wave, pt_[0], pt_[1], pt_[2], pt_[3], pt_[4], pt_[5], pt_[6], pt_[7]);
#endif
#undef P6_LOAD_E
#undef P6_LOAD_S
#undef P6_PREFETCH
}

template __global__ void k_pair_t6<0, 0>(const PairJob *, int, float *, const float *, const float *, float *, const u32 *, const u32 *, const u32 *,
                                      const u32 *, const float *, const float *, const float *, const float *const *, int);
template __global__ void k_pair_t6<1, 0>(const PairJob *, int, float *, const float *, const float *, float *, const u32 *, const u32 *, const u32 *,
                                      const u32 *, const float *, const float *, const float *, const float *const *, int);
