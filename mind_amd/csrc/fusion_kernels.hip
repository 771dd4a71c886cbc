// Symmetric relative-pose fusion (RelaFusionLayer x6) for gfx950 -- hand-written HIP, fp32 MFMA.
//
// Reference semantics: planners/mind/networks/network.py:165-232 (RelaFusionLayer),
// :259-268 (RelaFusionNet), :306-340 (FusionNet), planners/mind/utils.py:193-242 (RPE).
//
// Design (see DESIGN.md "k4"):
//  * one WAVE owns a column job (scene, j, i-range): 16 (i,j) pairs per tile, pair <-> lane&15.
//  * all per-pair 128x128 contractions run TRANSPOSED (features x pairs) on
//    v_mfma_f32_16x16x4_f32: A = weight fragment (LDS resident, pre-permuted on the host),
//    B = one f32 per lane.  With the K-slot permutation  slot(s,q) <-> feature 16*(s>>2)+4q+(s&3)
//    the C/D register layout of one GEMM *is* the B operand layout of the next, so
//    edge -> memory -> proj_edge chain through registers with no shuffles, and every LayerNorm is
//    an in-register reduction plus two cross-lane exchanges (lane^16, lane^32).
//  * proj_memory is rank-decomposed: W_m [e; x_j; x_i] = W_e e + S[j] + T[i] (S,T from k_token).
//  * K/V projections are folded algebraically: s = (W_k^T q).mem (+const), o = W_v (sum_i p mem) + b_v,
//    so only two 128x128 GEMMs per pair remain (W_e, W_p) instead of five.
//  * online softmax over i inside the wave; column partials (m, l, sum p*mem) go to k_token.
//  * the edge tensor is stored [scene][j][i][128] (query column major): a column job streams contiguous memory.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32 pk_bf16(float a, float b) {   // {bf16(a) in bits 0..15, bf16(b) in bits 16..31}, RNE
  f32x2_t v = {a, b};
  return __builtin_bit_cast(u32, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo_f32(u32 pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float bf_hi_f32(u32 pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }

#define D_ 128
#define PART_STRIDE 1040  // 8 (m) + 8 (l) + 8*128 (sum p*mem)

struct PairJob {
  long long edge_base;  // first pair index of the scene's edge tensor (pairs, not floats); pair (i, j) lives at edge_base + j * N + i
  int N;                // tokens in scene (a + l + 1)
  int j;                // column (query token)
  int t0, t1;           // i-tile range [t0, t1), 32 rows each
  int tok_base;         // first token of the scene in the flat token arrays
  int slot;             // partial-result slot
  int flags;            // bit0: column is an actor or the cls token
  int scene;
  long long edge_base_t;  // the same for the tile-native layout of k_pair_t (pair_tile_kernels.hip): columns padded to whole 16-row tiles
};

// per-layer small vectors, staged in LDS (floats): 7 x 128
//  0: gamma_m 1: beta_m 2: b_p 3: gamma_p 4: beta_p 5: gamma_e 6: beta_e
#define VT_GM 0
#define VT_BM 128
#define VT_BP 256
#define VT_GP 384
#define VT_BEP 512
#define VT_GE 640
#define VT_BE 768
#define VT_SIZE 896

// Eight waves (two per SIMD) share the two weight matrices: while one wave of a SIMD is in its LayerNorm / softmax /
// staging phases the other one keeps the MFMA busy.  Everything per-wave is sized so that the workgroup fits
// the 160 KB of LDS: the staging buffer holds a QUARTER tile (16 pairs x 32 features).
#define PAIR_WAVES 8
#define PAIR_THREADS (PAIR_WAVES * 64)
#define STAGE_FLOATS 512              // 16 rows x 8 chunks of 16 bytes
#define LDS_WAE 0
#define LDS_WAP 16384
#define LDS_STAGE 32768                               // 8 waves x 512 floats
#define LDS_PTAB (LDS_STAGE + PAIR_WAVES * STAGE_FLOATS)   // 8 waves x 128 floats
#define LDS_SVEC (LDS_PTAB + PAIR_WAVES * 128)        // 8 waves x 136 floats (S[j] + 8 softmax rescale factors)
#define LDS_VT (LDS_SVEC + PAIR_WAVES * 136)          // 896 floats
#define LDS_RT (LDS_VT + VT_SIZE)                     // rpe table 32 chunks x 8 x 4 = 1024 floats
#define LDS_TOTAL (LDS_RT + 1024)                     // 40896 floats = 163584 bytes (<= 163840)

#define LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// Hide a value's provenance from the optimiser: address arithmetic derived from it is recomputed
// where it is used (cheap VALU) instead of being hoisted out of the tile loop as hundreds of
// loop-invariant VGPRs (LDS offsets > 64 KB do not fit the DS immediate field).
#define OPAQUE(x) asm volatile("" : "+v"(x))

// Lane geometry (16 pairs per tile, v_mfma_f32_16x16x4_f32):
//   p = lane & 15 : pair (row i = i0 + p of column j),  q = lane >> 4 : feature quarter
//   a lane holds 8 x f32x4: chunk blk (0..7) = features 16*blk + 4*q + (0..3)  == the MFMA C/D layout
//   (col = lane&15, row = 4*(lane>>4) + reg) of output block blk, and == the B operand of k-group
//   blk (k-slot (s, q) <-> feature 16*(s>>2) + 4*q + (s&3)).
typedef f32x4 frag8[8];

__device__ __forceinline__ float red_quad(float v) {  // sum over the 4 lanes of a pair
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float red_max16(float v) {
  v = fmaxf(v, __shfl_xor(v, 8, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64));
  v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  return v;
}
__device__ __forceinline__ float red_sum16(float v) {
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}

// LayerNorm over the 128 features of each pair (two-pass, biased variance, eps 1e-5).
__device__ __forceinline__ void ln_pairs(frag8 &a, const float *vtq, int off_g, int off_b, bool relu) {
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int w = 0; w < 4; ++w) s += a[b][w];
  s = red_quad(s);
  const float mean = s * (1.0f / 128.0f);
  float v = 0.f;
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float d = a[b][w] - mean;
      v = fmaf(d, d, v);
    }
  v = red_quad(v);
  const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const f32x4 gm = *(const f32x4 *)(vtq + off_g + 16 * b);
    const f32x4 bt = *(const f32x4 *)(vtq + off_b + 16 * b);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float y = (a[b][w] - mean) * rstd * gm[w] + bt[w];
      if (relu) y = fmaxf(y, 0.f);
      a[b][w] = y;
    }
    if (b & 1) SCHED_FENCE();
  }
}

// acc[ob] += W_frag[ob] x B, B k-group s4 = bsrc[s4].  wa: LDS, packed [ob 8][s4 8][lane 64][4].
// 32 steps of 8 MFMAs (two output blocks x 4 k-steps, alternating accumulators so that dependent
// MFMAs are 64 cycles apart > the 40-cycle latency); the two A fragments of the next step are
// prefetched from LDS during the current one.  Scheduling fences bound the live ranges.
__device__ __forceinline__ void gemm128(frag8 &acc, const float *wa, const frag8 &bsrc, int lane) {
  // the lane offset is hidden as an INTEGER (not the pointer): the address space of `wa` (LDS) stays visible, so
  // the fragment reads are ds_read_b128 base+immediate instead of flat loads
  int lo = lane * 4;
  OPAQUE(lo);
  const float *wl = wa + lo;
  f32x4 c0 = *(const f32x4 *)(wl + (0 * 8 + 0) * 256);
  f32x4 c1 = *(const f32x4 *)(wl + (1 * 8 + 0) * 256);
#pragma unroll
  for (int st = 0; st < 32; ++st) {
    const int s4 = st >> 2, ob = (st & 3) * 2;
    f32x4 n0 = c0, n1 = c1;
    if (st < 31) {
      const int s4n = (st + 1) >> 2, obn = ((st + 1) & 3) * 2;
      n0 = *(const f32x4 *)(wl + (obn * 8 + s4n) * 256);
      n1 = *(const f32x4 *)(wl + ((obn + 1) * 8 + s4n) * 256);
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float bv = bsrc[s4][w];
      acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[w], bv, acc[ob], 0, 0, 0);
      acc[ob + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[w], bv, acc[ob + 1], 0, 0, 0);
    }
    c0 = n0; c1 = n1;
    SCHED_FENCE();
  }
}

// swizzled float offset of 16-byte chunk c (0..7) in a 128-byte staging row r (0..15): the 16 lanes of a
// ds_read_b128 pass (rows 0..15, one chunk each) and of a ds_write_b128 pass (2 rows x 8 chunks) hit 16
// distinct 16-byte bank groups
__device__ __forceinline__ int sw_pos(int r, int c) { return (r * 8 + (c ^ ((r >> 1) & 7))) * 4; }

// MODE 0: layer 0, edge built in-kernel from the relative pose encoding (no edge read)
// MODE 1: layers 1..5, edge read from HBM
// update_mode 0: update every edge; 1: update only flagged columns; 2: run only flagged columns,
// no edge update (last layer).
#ifdef MIND_PAIR_TRACE
#define PT_DECL long long pt_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long pt_t = clock64();
#define PT_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define PT(i) do { long long n_ = clock64(); pt_[i] += n_ - pt_t; pt_t = n_; } while (0)
#else
#define PT_DECL
#define PT_DRAIN()
#define PT(i)
#endif

template <int MODE>
__global__ __launch_bounds__(PAIR_THREADS, 1) void k_pair(const PairJob *__restrict__ jobs, int n_jobs,
                                                 float *__restrict__ edge, const float *__restrict__ ST,
                                                 const float *__restrict__ QK, float *__restrict__ part,
                                                 const float *__restrict__ WAe, const float *__restrict__ WAp,
                                                 const float *__restrict__ vtab, const float *__restrict__ rtab,
                                                 const float *__restrict__ tokpos,
                                                 const float *const *__restrict__ rpe_ptrs, int update_mode) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int p = lane & 15;
  const int q = lane >> 4;

  // ---- stage weights / tables (once per workgroup) ----
  {
    // all loads of a matrix in flight before the first LDS write (the load -> wait -> write loop serialised 16 L2 round trips)
    constexpr int PER = 4096 / PAIR_THREADS;
    f32x4 te[PER], tp[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) te[k] = ((const f32x4 *)WAe)[tid + k * PAIR_THREADS];
    if (update_mode != 2) {
#pragma unroll
      for (int k = 0; k < PER; ++k) tp[k] = ((const f32x4 *)WAp)[tid + k * PAIR_THREADS];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) ((f32x4 *)(lds + LDS_WAE))[tid + k * PAIR_THREADS] = te[k];
    if (update_mode != 2) {
#pragma unroll
      for (int k = 0; k < PER; ++k) ((f32x4 *)(lds + LDS_WAP))[tid + k * PAIR_THREADS] = tp[k];
    }
  }
  for (int i = tid; i < VT_SIZE; i += PAIR_THREADS) lds[LDS_VT + i] = vtab[i];
  if (MODE == 0)
    for (int i = tid; i < 1024; i += PAIR_THREADS) lds[LDS_RT + i] = rtab[i];
  __syncthreads();

  float *stage = lds + LDS_STAGE + wave * STAGE_FLOATS;
  float *ptab = lds + LDS_PTAB + wave * 128;
  float *svec = lds + LDS_SVEC + wave * 136;
  const float *vt = lds + LDS_VT;

  PT_DECL
  // wave-major job order: a launch with fewer jobs than wave slots spreads over all compute units (one wave per SIMD
  // first) instead of packing eight waves onto a few of them
  for (int job = wave * gridDim.x + blockIdx.x; job < n_jobs; job += gridDim.x * PAIR_WAVES) {
    const PairJob J = jobs[job];
    if (update_mode == 2 && !(J.flags & 1)) continue;
    if (J.t1 <= J.t0) continue;                  // padding job of the XCD-aware order
    const bool do_update = (update_mode == 0) || (update_mode == 1 && (J.flags & 1));
    const int N = J.N;
    const int j = J.j;
    LDS_FENCE();
    if (lane < 32) *(f32x4 *)(svec + lane * 4) = *(const f32x4 *)(ST + (size_t)(J.tok_base + j) * 256 + lane * 4);
    // this lane's 4 heads (lane quarter q < 2: heads 4q..4q+3; quarters 2,3 carry MFMA padding rows)
    float m_run[4], l_part[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { m_run[k] = -INFINITY; l_part[k] = 0.f; }
    // A fragment of the folded query (W_k^T q / 4): row = head (8 used of 16), k-slot = feature
    f32x4 qfrag[8];
    {
      const int hd = lane & 15;
      const float *qp = QK + (size_t)(J.tok_base + j) * 1024 + (hd & 7) * 128 + 4 * q;
#pragma unroll
      for (int s4 = 0; s4 < 8; ++s4) {
        f32x4 v = *(const f32x4 *)(qp + 16 * s4);
        if (hd >= 8) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        qfrag[s4] = v;
      }
    }
    float mbar[2][8];  // feature 64*hf + lane, head
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int k = 0; k < 8; ++k) mbar[hf][k] = 0.f;
    float pj[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
      const f32x4 t = *(const f32x4 *)(tokpos + (size_t)(J.tok_base + j) * 4);
      pj[0] = t[0]; pj[1] = t[1]; pj[2] = t[2]; pj[3] = t[3];
    }
    LDS_FENCE();

    for (int tile = J.t0; tile < J.t1; ++tile) {
      const int i0 = tile * 16;
      int lq = q, lp = p, ll = lane;
      OPAQUE(lq); OPAQUE(lp); OPAQUE(ll);
      const float *vtq = vt + lq * 4;
      const int i = i0 + p;
      const bool valid = i < N;
      const int ic = valid ? i : (N - 1);
      frag8 ef;
      if (MODE == 1) {
        // ---- edge tile -> registers through the swizzled staging buffer: 4 x (16 rows x 128 B),
        //      every global access is a full 128-byte row segment (coalesced), LDS accesses conflict-free
        //      all eight global loads of the tile are issued before the first LDS pass (one HBM round trip per tile)
        f32x4 raw[8];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            int irow = i0 + 8 * n + (ll >> 3);
            irow = irow < N ? irow : N - 1;
            raw[2 * qt + n] = *(const f32x4 *)(edge + (((size_t)J.edge_base + (size_t)j * N + irow) << 7) + qt * 32 + (ll & 7) * 4);
          }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
          for (int n = 0; n < 2; ++n) *(f32x4 *)(stage + sw_pos(8 * n + (ll >> 3), ll & 7)) = raw[2 * qt + n];
          LDS_FENCE();
#pragma unroll
          for (int b2 = 0; b2 < 2; ++b2) ef[2 * qt + b2] = *(const f32x4 *)(stage + sw_pos(lp, 4 * b2 + lq));
          LDS_FENCE();
        }
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
        const float *rt = lds + LDS_RT + lq * 32;
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

      PT(0);
      // ---- memory = ReLU(LN(W_e e + S[j] + T[i]))   (network.py:197-199, rank-decomposed)
      frag8 mem;
#pragma unroll
      for (int b = 0; b < 8; ++b) mem[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
      gemm128(mem, lds + LDS_WAE, ef, lane);
      PT(1);
      {
        const float *Ti = ST + (size_t)(J.tok_base + ic) * 256 + 128 + lq * 4;
        const float *svq = svec + lq * 4;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const f32x4 t4 = *(const f32x4 *)(Ti + 16 * b);
          const f32x4 s4 = *(const f32x4 *)(svq + 16 * b);
          mem[b] += t4 + s4;
          if ((b & 3) == 3) SCHED_FENCE();
        }
      }
      ln_pairs(mem, vtq, VT_GM, VT_BM, true);
      SCHED_FENCE();
      PT(2);

      PT(5);
      // ---- edge update e' = LN_e(e + ReLU(LN(W_p mem + b_p)))   (network.py:201-202)
      if (do_update) {
        frag8 up;
#pragma unroll
        for (int b = 0; b < 8; ++b) up[b] = *(const f32x4 *)(vtq + VT_BP + 16 * b);
        gemm128(up, lds + LDS_WAP, mem, lane);
        PT(6);
        ln_pairs(up, vtq, VT_GP, VT_BEP, true);
#pragma unroll
        for (int b = 0; b < 8; ++b) up[b] += ef[b];
        ln_pairs(up, vtq, VT_GE, VT_BE, false);
        // store through the staging buffer: full 128-byte row segments
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
          for (int b2 = 0; b2 < 2; ++b2) *(f32x4 *)(stage + sw_pos(lp, 4 * b2 + lq)) = up[2 * qt + b2];
          LDS_FENCE();
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const int r = 8 * n + (ll >> 3);
            const int cp = ll & 7;
            const f32x4 v = *(const f32x4 *)(stage + sw_pos(r, cp));
            if (i0 + r < N)
              *(f32x4 *)(edge + (((size_t)J.edge_base + (size_t)j * N + (i0 + r)) << 7) + qt * 32 + cp * 4) = v;
          }
          LDS_FENCE();
        }
      }
      SCHED_FENCE();
      // ---- attention scores on the MFMA: S^T[head, pair] = QK_j[head, :] . mem^T[:, pair]; the memory
      //      tile is already the B operand (chained layout).  Two accumulators hide the 40-cycle latency.
      f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f}, sb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 8; s4 += 2)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          sa = __builtin_amdgcn_mfma_f32_16x16x4f32(qfrag[s4][w], mem[s4][w], sa, 0, 0, 0);
          sb = __builtin_amdgcn_mfma_f32_16x16x4f32(qfrag[s4 + 1][w], mem[s4 + 1][w], sb, 0, 0, 0);
        }
      sa += sb;       // lane (p, q): rows 4q..4q+3 = heads 4q..4q+3 of pair p (q < 2)
      PT(3);
      // ---- online softmax over i: lanes of quarter q own heads 4q..4q+3
      float pr[4], scl[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sv = valid ? sa[r] : -INFINITY;
        const float mx = red_max16(sv);
        const float m_new = fmaxf(m_run[r], mx);
        scl[r] = __expf(m_run[r] - m_new);
        pr[r] = valid ? __expf(sv - m_new) : 0.f;
        l_part[r] = l_part[r] * scl[r] + pr[r];
        m_run[r] = m_new;
      }
      LDS_FENCE();
      if (q < 2) {
        f32x4 a;
        a[0] = pr[0]; a[1] = pr[1]; a[2] = pr[2]; a[3] = pr[3];
        *(f32x4 *)(ptab + p * 8 + 4 * q) = a;
        if (p == 0) {
          f32x4 b;
          b[0] = scl[0]; b[1] = scl[1]; b[2] = scl[2]; b[3] = scl[3];
          *(f32x4 *)(svec + 128 + 4 * q) = b;        // rescale factors, broadcast through LDS
        }
      }
      LDS_FENCE();
      {
        const f32x4 s0 = *(const f32x4 *)(svec + 128), s1 = *(const f32x4 *)(svec + 132);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          mbar[0][k] *= s0[k]; mbar[1][k] *= s0[k];
          mbar[0][k + 4] *= s1[k]; mbar[1][k + 4] *= s1[k];
        }
      }
      PT(4);
      // ---- mbar[hd][f] += sum_pairs p[pair][hd] * mem[pair][f]  (V projection folded out),
      //      transposing mem through the staging buffer one 32-feature quarter at a time: quarter qt holds the
      //      features 64*(qt>>1) + lane of the lanes with (lane >> 5) == (qt & 1); the other half-wave adds zeros
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        const int hf = qt >> 1;
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) *(f32x4 *)(stage + sw_pos(lp, 4 * b2 + lq)) = mem[2 * qt + b2];
        LDS_FENCE();
        const bool mine = (ll >> 5) == (qt & 1);
        const int fq = ll & 31;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          float val = stage[sw_pos(t, fq >> 2) + (fq & 3)];
          val = mine ? val : 0.f;
          const f32x4 pa = *(const f32x4 *)(ptab + t * 8);
          const f32x4 pb = *(const f32x4 *)(ptab + t * 8 + 4);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            mbar[hf][k] = fmaf(pa[k], val, mbar[hf][k]);
            mbar[hf][k + 4] = fmaf(pb[k], val, mbar[hf][k + 4]);
          }
          if ((t & 3) == 3) SCHED_FENCE();
        }
        LDS_FENCE();
      }

      PT(7);
    }  // tiles

    // ---- column partial: m[8], l[8], mbar[8][128]
    float *po = part + (size_t)J.slot * PART_STRIDE;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float l = red_sum16(l_part[r]);
      if (p == 0 && q < 2) { po[4 * q + r] = m_run[r]; po[8 + 4 * q + r] = l; }
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int hd = 0; hd < 8; ++hd) po[16 + hd * 128 + hf * 64 + lane] = mbar[hf][hd];
  }
#ifdef MIND_PAIR_TRACE
  if (blockIdx.x == 0 && tid == 0)
    printf("[k_pair<%d> um=%d] cycles: load %lld gemm1 %lld T+LN1 %lld scores %lld softmax %lld mbar %lld gemm2 %lld LN23+store %lld\n", MODE,
           update_mode, pt_[0], pt_[1], pt_[2], pt_[3], pt_[4], pt_[5], pt_[6], pt_[7]);
#endif
}

template __global__ void k_pair<0>(const PairJob *, int, float *, const float *, const float *, float *,
                                   const float *, const float *, const float *, const float *,
                                   const float *, const float *const *, int);
template __global__ void k_pair<1>(const PairJob *, int, float *, const float *, const float *, float *,
                                   const float *, const float *, const float *, const float *,
                                   const float *, const float *const *, int);

extern "C" size_t mind_pair_lds_bytes() { return (size_t)LDS_TOTAL * sizeof(float); }

// =================================================================================================
// k_token: per-token epilogue of layer L (combine column partials, V/out projections, LN, FFN, LN;
// network.py:177-179,222-232) and prologue of layer L+1 (S, T, q, qk).  TPW tokens per workgroup,
// 512 threads = 4 k-groups x 128 output features; weights are stored transposed [in][out] so that a
// wave reads 64 consecutive floats per k.
// =================================================================================================
#define TOK_TPW_BIG 8      // tokens per k_token workgroup: big batches
#define TOK_TPW_SMALL 4    // ... small batches (same arithmetic, see k_token)

struct TokMeta {
  int type;      // 0 actor, 1 lane, 2 cls
  int src;       // row in actor_feat / lane_feat
  int slot0;     // first partial slot of this token's column
  int nsplit;    // number of partial slots
  int flags;     // bit0: actor or cls (needed by the last layer)
  int pad0, pad1, pad2;
};

struct TokWeights {
  // epilogue of layer L (may be null when init)
  const float *WvT, *bv, *WoT, *bo, *g2, *b2, *W1T, *b1, *W2T, *bb2, *g3, *b3;
  // prologue of layer L+1 (may be null for the last layer)
  const float *WsT, *WtT, *bm, *WqT, *bq, *Wk;
  // init projections
  const float *WpaT, *bpa, *gpa, *bepa, *WplT, *bpl, *gpl, *bepl;
};

// ---- k_token building blocks.  512 threads = 4 k-groups x 128 output features: a projection's reduction
// dimension is split four ways so that all of its weight loads are in flight at once (the kernel runs a few
// dozen workgroups whose weights arrive cold from the Infinity Cache: it is bound by those round trips, not
// by FLOPs); the partial sums meet in LDS, after which thread (kg, col) owns tokens 2kg, 2kg+1 of feature col.
#define TKG 4
#define TT_THREADS (TKG * 128)

// partial y[t] = sum_{k in this group's quarter} WT[k*ldo + col] * xin[t][k]   (K/4 <= 64 weights in registers)
template <int K, int TP>
__device__ __forceinline__ void matvec_part(float (&acc)[TP], const float *__restrict__ WT, int ldo, int col,
                                            const float *xin, int ldx, int kg) {
  constexpr int KQ = K / TKG;
  const int k0 = kg * KQ;
  float w[KQ];
#pragma unroll
  for (int k = 0; k < KQ; ++k) w[k] = WT[(size_t)(k0 + k) * ldo + col];
#pragma unroll
  for (int t = 0; t < TP; ++t) acc[t] = 0.f;
#pragma unroll
  for (int k = 0; k < KQ; k += 4)
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const float4 xv = *reinterpret_cast<const float4 *>(xin + t * ldx + k0 + k);
      acc[t] = fmaf(w[k], xv.x, acc[t]);
      acc[t] = fmaf(w[k + 1], xv.y, acc[t]);
      acc[t] = fmaf(w[k + 2], xv.z, acc[t]);
      acc[t] = fmaf(w[k + 3], xv.w, acc[t]);
    }
}

// the two halves of matvec_part (k_token_m issues the weight loads of INDEPENDENT projections together and exchanges their partial sums behind
// one pair of barriers: same loads, same multiply-adds, same summation order)
template <int K>
__device__ __forceinline__ void mv_load(float (&w)[K / TKG], const float *__restrict__ WT, int ldo, int col, int kg) {
  constexpr int KQ = K / TKG;
  const int k0 = kg * KQ;
#pragma unroll
  for (int k = 0; k < KQ; ++k) w[k] = WT[(size_t)(k0 + k) * ldo + col];
}
template <int K, int TP>
__device__ __forceinline__ void mv_fma(float (&acc)[TP], const float (&w)[K / TKG], const float *xin, int ldx, int kg) {
  constexpr int KQ = K / TKG;
  const int k0 = kg * KQ;
#pragma unroll
  for (int t = 0; t < TP; ++t) acc[t] = 0.f;
#pragma unroll
  for (int k = 0; k < KQ; k += 4)
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const float4 xv = *reinterpret_cast<const float4 *>(xin + t * ldx + k0 + k);
      acc[t] = fmaf(w[k], xv.x, acc[t]);
      acc[t] = fmaf(w[k + 1], xv.y, acc[t]);
      acc[t] = fmaf(w[k + 2], xv.z, acc[t]);
      acc[t] = fmaf(w[k + 3], xv.w, acc[t]);
    }
}
// N independent projections' partials meet behind ONE pair of barriers (rk holds N x [TKG][TP][128]); per projection the sums are ksum's
template <int TP, int N>
__device__ __forceinline__ void ksum_n(const float (&acc)[N][TP], const float (&bias)[N], float *rk, int kg, int col, float (&r0)[N], float (&r1)[N]) {
  constexpr int TPW = TP, ST = TKG * TP * 128;
  __syncthreads();
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int t = 0; t < TPW; ++t) rk[n * ST + (kg * TPW + t) * 128 + col] = acc[n][t];
  __syncthreads();
  const int t0 = (TP / TKG) * kg;
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const float *q = rk + n * ST;
    r0[n] = ((q[(0 * TPW + t0) * 128 + col] + q[(1 * TPW + t0) * 128 + col]) + (q[(2 * TPW + t0) * 128 + col] + q[(3 * TPW + t0) * 128 + col])) + bias[n];
    r1[n] = 0.f;
    if (TP / TKG == 2)
      r1[n] = ((q[(0 * TPW + t0 + 1) * 128 + col] + q[(1 * TPW + t0 + 1) * 128 + col]) + (q[(2 * TPW + t0 + 1) * 128 + col] + q[(3 * TPW + t0 + 1) * 128 + col])) + bias[n];
  }
}

// meet the four partials: on return r0/r1 = bias + full sums for this thread's tokens (fixed order): 2kg, 2kg+1 of a workgroup of
// eight tokens, kg alone (r1 unused) of a workgroup of four.  Contains two barriers; rk is [TKG][TP][128].
template <int TP>
__device__ __forceinline__ void ksum(const float (&acc)[TP], float bias, float *rk, int kg, int col, float &r0, float &r1) {
  constexpr int TPW = TP;
  __syncthreads();                      // previous users of rk are done
#pragma unroll
  for (int t = 0; t < TPW; ++t) rk[(kg * TPW + t) * 128 + col] = acc[t];
  __syncthreads();
  const int t0 = (TP / TKG) * kg;
  r0 = ((rk[(0 * TPW + t0) * 128 + col] + rk[(1 * TPW + t0) * 128 + col]) +
        (rk[(2 * TPW + t0) * 128 + col] + rk[(3 * TPW + t0) * 128 + col])) + bias;
  r1 = 0.f;
  if (TP / TKG == 2)
    r1 = ((rk[(0 * TPW + t0 + 1) * 128 + col] + rk[(1 * TPW + t0 + 1) * 128 + col]) +
          (rk[(2 * TPW + t0 + 1) * 128 + col] + rk[(3 * TPW + t0 + 1) * 128 + col])) + bias;
}

// sum over the 128 features of this k-group's two tokens (two waves per group); rd is [TKG][2][2]
__device__ __forceinline__ void gsum2(float &a, float &b, float *rd, int kg, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
  const int wv = (tid >> 6) & 1;
  __syncthreads();
  if ((tid & 63) == 0) { rd[(kg * 2 + wv) * 2] = a; rd[(kg * 2 + wv) * 2 + 1] = b; }
  __syncthreads();
  a = rd[(kg * 2) * 2] + rd[(kg * 2 + 1) * 2];
  b = rd[(kg * 2) * 2 + 1] + rd[(kg * 2 + 1) * 2 + 1];
}

// LayerNorm over the 128 features of the group's two tokens (eps 1e-5), in place
__device__ __forceinline__ void ln2tok(float &p0, float &p1, float g, float be, float *rd, int kg, int tid) {
  float s0 = p0, s1 = p1;
  gsum2(s0, s1, rd, kg, tid);
  const float d0 = p0 - s0 * (1.0f / 128.0f), d1 = p1 - s1 * (1.0f / 128.0f);
  s0 = d0 * d0; s1 = d1 * d1;
  gsum2(s0, s1, rd, kg, tid);
  p0 = d0 * (1.0f / sqrtf(s0 * (1.0f / 128.0f) + 1e-5f)) * g + be;
  p1 = d1 * (1.0f / sqrtf(s1 * (1.0f / 128.0f) + 1e-5f)) * g + be;
}

// mode bits: 1 = init (x0 from actor/lane features), 2 = has epilogue, 4 = has prologue, 8 = only flagged,
// 16 = write the folded query as bf16 hi / lo A fragments (for k_pair_bf) instead of fp32
// TPW = 8 tokens per workgroup (big batches: every weight fetched from L2 serves eight tokens) or 4 (small batches: twice the
// workgroups, half the LDS operand traffic in each -- the per-stage time of a lone scene's launch).  The k split and every
// summation order are the same, so the two instantiations give bit-identical results.
// MG (k_token_m, small launches): the projections that do not depend on each other -- the two halves of the feed-forward layer's first
// matrix, and S / T / q of the next layer's prologue -- load their weights together and exchange their partial sums behind one pair of
// barriers (ksum_n): three of the launch's ten dependent stages fewer, every sum in the same order: bit-identical to the plain form.
template <int TPW, bool MG>
__device__ __forceinline__ void k_token_body(const TokMeta *__restrict__ meta, int n_tok, int mode,
                                             const float *__restrict__ actor_feat,
                                             const float *__restrict__ lane_feat, float *__restrict__ x,
                                             const float *__restrict__ part, float *__restrict__ ST,
                                             float *__restrict__ QK, const TokWeights &W) {
  __shared__ __attribute__((aligned(16))) float xs[TPW][132];        // current token vectors
  __shared__ __attribute__((aligned(16))) float tmp[TPW][260];       // scratch (o / h1 up to 256 wide)
  __shared__ __attribute__((aligned(16))) float mb[TPW][8][132];     // normalised sum p*mem per head
  __shared__ float rk[(MG ? 3 : 1) * TKG * TPW * 128];               // k-group partial sums (MG: of up to three projections at once)
  __shared__ float rd[TKG * 2 * 2];
  __shared__ float cw[TPW][8][8];       // combine weights per (token, head, split<=8)
  const int tid = threadIdx.x, kg = tid >> 7, col = tid & 127;
  const int tok0 = blockIdx.x * TPW;
  const int nt = min(TPW, n_tok - tok0);
  constexpr bool TWO = TPW / TKG == 2;
  const int ta = (TPW / TKG) * kg, tb = TWO ? ta + 1 : ta;          // the token(s) this thread owns after a ksum
  const bool va = ta < nt, vb = TWO && tb < nt;
#ifdef MIND_TOKEN_TRACE
  long long tt_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tq_ = clock64();
#define TT(i) do { const long long n_ = clock64(); tt_[i] += n_ - tq_; tq_ = n_; } while (0)
#else
#define TT(i) do {} while (0)
#endif

  // ---- load x (or build x0)
  if (mode & 1) {
    // FusionNet projections (network.py:313-314, 323-324): Linear + LN + ReLU, cls token = zeros
    int tya = 2, tyb = 2;
    float fa = 0.f, fb = 0.f;
    if (va) {
      const TokMeta m = meta[tok0 + ta];
      tya = m.type;
      if (m.type == 0) fa = actor_feat[(size_t)m.src * 128 + col];
      else if (m.type == 1) fa = lane_feat[(size_t)m.src * 128 + col];
    }
    if (vb) {
      const TokMeta m = meta[tok0 + tb];
      tyb = m.type;
      if (m.type == 0) fb = actor_feat[(size_t)m.src * 128 + col];
      else if (m.type == 1) fb = lane_feat[(size_t)m.src * 128 + col];
    }
    tmp[ta][col] = fa; if (TWO) tmp[tb][col] = fb;
    __syncthreads();
    float acc[TPW], a0, a1, l0, l1;
    matvec_part<128, TPW>(acc, W.WpaT, 128, col, &tmp[0][0], 260, kg);
    ksum<TPW>(acc, W.bpa[col], rk, kg, col, a0, a1);
    matvec_part<128, TPW>(acc, W.WplT, 128, col, &tmp[0][0], 260, kg);
    ksum<TPW>(acc, W.bpl[col], rk, kg, col, l0, l1);
    float p0 = tya == 0 ? a0 : l0, p1 = tyb == 0 ? a1 : l1;
    {
      // LN with per-token (type-dependent) affine: normalise first, then scale
      float s0 = p0, s1 = p1;
      gsum2(s0, s1, rd, kg, tid);
      const float d0 = p0 - s0 * (1.0f / 128.0f), d1 = p1 - s1 * (1.0f / 128.0f);
      s0 = d0 * d0; s1 = d1 * d1;
      gsum2(s0, s1, rd, kg, tid);
      const float ga = W.gpa[col], gl = W.gpl[col], ba = W.bepa[col], bl = W.bepl[col];
      p0 = fmaxf(d0 * (1.0f / sqrtf(s0 * (1.0f / 128.0f) + 1e-5f)) * (tya == 0 ? ga : gl) + (tya == 0 ? ba : bl), 0.f);
      p1 = fmaxf(d1 * (1.0f / sqrtf(s1 * (1.0f / 128.0f) + 1e-5f)) * (tyb == 0 ? ga : gl) + (tyb == 0 ? ba : bl), 0.f);
      if (tya == 2) p0 = 0.f;
      if (tyb == 2) p1 = 0.f;
    }
    xs[ta][col] = p0; if (TWO) xs[tb][col] = p1;
    __syncthreads();
  } else {
    xs[ta][col] = va ? x[(size_t)(tok0 + ta) * 128 + col] : 0.f;
    if (TWO) xs[tb][col] = vb ? x[(size_t)(tok0 + tb) * 128 + col] : 0.f;
    __syncthreads();
  }
  TT(0);
  if (mode & 2) {
    // ---- combine split partials: weights exp(m_s - M) / L  (softmax over i finished here)
    if (tid < TPW * 8) {
      const int t = tid >> 3, hd = tid & 7;
      float wgt[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) wgt[s] = 0.f;
      if (t < nt) {
        const TokMeta m = meta[tok0 + t];
        if (!((mode & 8) && !(m.flags & 1))) {
          float M = -INFINITY;
          for (int s = 0; s < m.nsplit; ++s) M = fmaxf(M, part[(size_t)(m.slot0 + s) * PART_STRIDE + hd]);
          float L = 0.f;
          for (int s = 0; s < m.nsplit; ++s) {
            const float *ps = part + (size_t)(m.slot0 + s) * PART_STRIDE;
            const float e = expf(ps[hd] - M);
            wgt[s] = e;
            L += e * ps[8 + hd];
          }
          const float inv = 1.0f / L;
          for (int s = 0; s < m.nsplit; ++s) wgt[s] *= inv;
        }
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) cw[t][hd][s] = wgt[s];
    }
    __syncthreads();
    TT(1);
#pragma unroll
    for (int q = 0; q < TPW / TKG; ++q) {
      const int t = ta + q;
      int ns = 0, slot0 = 0;
      if (t < nt) {
        const TokMeta m = meta[tok0 + t];
        ns = ((mode & 8) && !(m.flags & 1)) ? 0 : m.nsplit;
        slot0 = m.slot0;
      }
      // all 8 heads' loads of a split are issued together (the split loop has a run-time trip count)
      float v[8];
#pragma unroll
      for (int hd = 0; hd < 8; ++hd) v[hd] = 0.f;
      // two splits' loads in flight at a time (the partials of a big batch come from HBM: one round trip per split was the longest
      // stage of a big launch, MIND_TOKEN_TRACE); the accumulation order over the splits is unchanged
      int s = 0;
      for (; s + 2 <= ns; s += 2) {
        const float *ps = part + (size_t)(slot0 + s) * PART_STRIDE + 16 + col;
        float pv[8], pw[8];
#pragma unroll
        for (int hd = 0; hd < 8; ++hd) { pv[hd] = ps[hd * 128]; pw[hd] = ps[PART_STRIDE + hd * 128]; }
#pragma unroll
        for (int hd = 0; hd < 8; ++hd) v[hd] = fmaf(cw[t][hd][s], pv[hd], v[hd]);
#pragma unroll
        for (int hd = 0; hd < 8; ++hd) v[hd] = fmaf(cw[t][hd][s + 1], pw[hd], v[hd]);
      }
      if (s < ns) {
        const float *ps = part + (size_t)(slot0 + s) * PART_STRIDE + 16 + col;
        float pv[8];
#pragma unroll
        for (int hd = 0; hd < 8; ++hd) pv[hd] = ps[hd * 128];
#pragma unroll
        for (int hd = 0; hd < 8; ++hd) v[hd] = fmaf(cw[t][hd][s], pv[hd], v[hd]);
      }
#pragma unroll
      for (int hd = 0; hd < 8; ++hd) mb[t][hd][col] = v[hd];
    }
    __syncthreads();
    TT(2);
    // ---- o = W_v,h mbar_h + b_v  (output feature f = hd*16+d uses head hd = f>>4)
    {
      float acc[TPW], o0, o1;
      matvec_part<128, TPW>(acc, W.WvT, 128, col, &mb[0][col >> 4][0], 8 * 132, kg);
      ksum<TPW>(acc, W.bv[col], rk, kg, col, o0, o1);
      tmp[ta][col] = o0; if (TWO) tmp[tb][col] = o1;
      __syncthreads();
    }
    TT(3);
    // ---- att = W_o o + b_o ; x1 = LN2(x + att)
    {
      float acc[TPW], p0, p1;
      matvec_part<128, TPW>(acc, W.WoT, 128, col, &tmp[0][0], 260, kg);
      ksum<TPW>(acc, W.bo[col], rk, kg, col, p0, p1);
      p0 += xs[ta][col]; p1 += xs[tb][col];
      ln2tok(p0, p1, W.g2[col], W.b2[col], rd, kg, tid);
      __syncthreads();                  // every k-group has finished reading xs
      xs[ta][col] = p0; if (TWO) xs[tb][col] = p1;
      __syncthreads();
    }
    TT(4);
    // ---- FFN 128 -> 256 -> 128, x2 = LN3(x1 + ff)
    {
      float acc[TPW], h0, h1;
      if (MG) {
        float wa[32], wb[32], ac2[2][TPW], r0[2], r1[2];
        mv_load<128>(wa, W.W1T, 256, col, kg);
        mv_load<128>(wb, W.W1T, 256, 128 + col, kg);
        mv_fma<128, TPW>(ac2[0], wa, &xs[0][0], 132, kg);
        mv_fma<128, TPW>(ac2[1], wb, &xs[0][0], 132, kg);
        const float bs[2] = {W.b1[col], W.b1[128 + col]};
        ksum_n<TPW, 2>(ac2, bs, rk, kg, col, r0, r1);
        tmp[ta][col] = fmaxf(r0[0], 0.f); if (TWO) tmp[tb][col] = fmaxf(r1[0], 0.f);
        tmp[ta][128 + col] = fmaxf(r0[1], 0.f); if (TWO) tmp[tb][128 + col] = fmaxf(r1[1], 0.f);
      } else {
        matvec_part<128, TPW>(acc, W.W1T, 256, col, &xs[0][0], 132, kg);
        ksum<TPW>(acc, W.b1[col], rk, kg, col, h0, h1);
        tmp[ta][col] = fmaxf(h0, 0.f); if (TWO) tmp[tb][col] = fmaxf(h1, 0.f);
        matvec_part<128, TPW>(acc, W.W1T, 256, 128 + col, &xs[0][0], 132, kg);
        ksum<TPW>(acc, W.b1[128 + col], rk, kg, col, h0, h1);
        tmp[ta][128 + col] = fmaxf(h0, 0.f); if (TWO) tmp[tb][128 + col] = fmaxf(h1, 0.f);
      }
      __syncthreads();
      float p0, p1;
      matvec_part<256, TPW>(acc, W.W2T, 128, col, &tmp[0][0], 260, kg);
      ksum<TPW>(acc, W.bb2[col], rk, kg, col, p0, p1);
      p0 += xs[ta][col]; p1 += xs[tb][col];
      ln2tok(p0, p1, W.g3[col], W.b3[col], rd, kg, tid);
      __syncthreads();
      xs[ta][col] = p0; if (TWO) xs[tb][col] = p1;
      __syncthreads();
    }
  }
  TT(5);
  // ---- write x
  if (va) x[(size_t)(tok0 + ta) * 128 + col] = xs[ta][col];
  if (vb) x[(size_t)(tok0 + tb) * 128 + col] = xs[tb][col];

  if (mode & 4) {
    // ---- prologue of the next layer: S = W_s x, T = W_t x + b_m, q = W_q x + b_q,
    //      qk[hd][f] = sum_d q[hd*16+d] W_k[hd*16+d][f] / 4      (scale 1/sqrt(16))
    float acc[TPW], r0, r1;
    if (MG) {
      float ws[32], wt[32], wq[32], ac3[3][TPW], q0[3], q1[3];
      mv_load<128>(ws, W.WsT, 128, col, kg);
      mv_load<128>(wt, W.WtT, 128, col, kg);
      mv_load<128>(wq, W.WqT, 128, col, kg);
      mv_fma<128, TPW>(ac3[0], ws, &xs[0][0], 132, kg);
      mv_fma<128, TPW>(ac3[1], wt, &xs[0][0], 132, kg);
      mv_fma<128, TPW>(ac3[2], wq, &xs[0][0], 132, kg);
      const float bs[3] = {0.f, W.bm[col], W.bq[col]};
      ksum_n<TPW, 3>(ac3, bs, rk, kg, col, q0, q1);
      if (va) { ST[(size_t)(tok0 + ta) * 256 + col] = q0[0]; ST[(size_t)(tok0 + ta) * 256 + 128 + col] = q0[1]; }
      if (vb) { ST[(size_t)(tok0 + tb) * 256 + col] = q1[0]; ST[(size_t)(tok0 + tb) * 256 + 128 + col] = q1[1]; }
      tmp[ta][col] = q0[2]; if (TWO) tmp[tb][col] = q1[2];
    } else {
      matvec_part<128, TPW>(acc, W.WsT, 128, col, &xs[0][0], 132, kg);
      ksum<TPW>(acc, 0.f, rk, kg, col, r0, r1);
      if (va) ST[(size_t)(tok0 + ta) * 256 + col] = r0;
      if (vb) ST[(size_t)(tok0 + tb) * 256 + col] = r1;
      matvec_part<128, TPW>(acc, W.WtT, 128, col, &xs[0][0], 132, kg);
      ksum<TPW>(acc, W.bm[col], rk, kg, col, r0, r1);
      if (va) ST[(size_t)(tok0 + ta) * 256 + 128 + col] = r0;
      if (vb) ST[(size_t)(tok0 + tb) * 256 + 128 + col] = r1;
      matvec_part<128, TPW>(acc, W.WqT, 128, col, &xs[0][0], 132, kg);
      ksum<TPW>(acc, W.bq[col], rk, kg, col, r0, r1);
      tmp[ta][col] = r0; if (TWO) tmp[tb][col] = r1;
    }
    __syncthreads();
    TT(6);
    // each k-group takes two of the eight heads, all tokens
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int hd = 2 * kg + hh;
      float w[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) w[d] = W.Wk[(size_t)(hd * 16 + d) * 128 + col];
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = 0.f;
#pragma unroll
      for (int d = 0; d < 16; d += 4)
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const float4 xv = *reinterpret_cast<const float4 *>(&tmp[t][hd * 16 + d]);
          acc[t] = fmaf(w[d], xv.x, acc[t]);
          acc[t] = fmaf(w[d + 1], xv.y, acc[t]);
          acc[t] = fmaf(w[d + 2], xv.z, acc[t]);
          acc[t] = fmaf(w[d + 3], xv.w, acc[t]);
        }
      if (mode & 16) {
        // bf16 hi / lo parts in the A-operand order of k_pair_bf (pair_bf16_kernels.hip): per token 2048 x u16 =
        // [part 2][k-group 4][row 32 = head 8 x lane quarter 4][slot 8], feature 16 (2g + (i >> 2)) + 4q + (i & 3) <-> slot i
        const int g = col >> 5, qq = (col >> 2) & 3, i = 4 * ((col >> 4) & 1) + (col & 3);
        const int idx = ((g * 32) + hd * 4 + qq) * 8 + i;
        // mode bit 32 (k_pair_t6): three parts hi / mid / lo (the exact split of the fp32 value), 1536 dwords per token
        const size_t qks = (mode & 32) ? 1536 : 1024;
        for (int t = 0; t < nt; ++t) {
          const float v = acc[t] * 0.25f;
          const u32 h = pk_bf16(v, v) & 0xffffu;
          const float r1 = v - bf_lo_f32(h);
          const u32 l = pk_bf16(r1, 0.f) & 0xffffu;
          unsigned short *qs = reinterpret_cast<unsigned short *>(QK + (size_t)(tok0 + t) * qks);
          qs[idx] = (unsigned short)h;
          qs[1024 + idx] = (unsigned short)l;
          if (mode & 32) qs[2048 + idx] = (unsigned short)(pk_bf16(r1 - bf_lo_f32(l), 0.f) & 0xffffu);
        }
      } else {
        for (int t = 0; t < nt; ++t) QK[(size_t)(tok0 + t) * 1024 + hd * 128 + col] = acc[t] * 0.25f;
      }
    }
  }
  TT(7);
#ifdef MIND_TOKEN_TRACE
  if (blockIdx.x == 0 && tid == 0)
    printf("[k_token mode %d ntok %d] cycles: load %lld cw %lld combine %lld Wv %lld Wo+LN2 %lld FFN+LN3 %lld S,T,q %lld QK %lld\n", mode, n_tok,
           tt_[0], tt_[1], tt_[2], tt_[3], tt_[4], tt_[5], tt_[6], tt_[7]);
#endif
}

template <int TPW>
__global__ __launch_bounds__(TT_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_token(const TokMeta *__restrict__ meta, int n_tok, int mode,
                                                      const float *__restrict__ actor_feat,
                                                      const float *__restrict__ lane_feat, float *__restrict__ x,
                                                      const float *__restrict__ part, float *__restrict__ ST,
                                                      float *__restrict__ QK, TokWeights W) {
  k_token_body<TPW, false>(meta, n_tok, mode, actor_feat, lane_feat, x, part, ST, QK, W);
}
// small launches (a lone demo-size scene: 24 workgroups, one per CU, two waves per SIMD): independent projections merged
__global__ __launch_bounds__(TT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_token_m(const TokMeta *__restrict__ meta, int n_tok, int mode,
                                                      const float *__restrict__ actor_feat,
                                                      const float *__restrict__ lane_feat, float *__restrict__ x,
                                                      const float *__restrict__ part, float *__restrict__ ST,
                                                      float *__restrict__ QK, TokWeights W) {
  k_token_body<TOK_TPW_SMALL, true>(meta, n_tok, mode, actor_feat, lane_feat, x, part, ST, QK, W);
}
