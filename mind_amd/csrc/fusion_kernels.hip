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
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define D_ 128
#define PART_STRIDE 1040  // 8 (m) + 8 (l) + 8*128 (sum p*mem)

struct PairJob {
  long long edge_base;  // first pair index of the scene's edge tensor (pairs, not floats)
  int N;                // tokens in scene (a + l + 1)
  int j;                // column (query token)
  int t0, t1;           // i-tile range [t0, t1), 32 rows each
  int tok_base;         // first token of the scene in the flat token arrays
  int slot;             // partial-result slot
  int flags;            // bit0: column is an actor or the cls token
  int scene;
  int pad0, pad1;
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

#define LDS_WAE 0
#define LDS_WAP 16384
#define LDS_STAGE 32768               // 4 waves x 1024 floats
#define LDS_PTAB (LDS_STAGE + 4096)   // 4 waves x 128 floats
#define LDS_SVEC (LDS_PTAB + 512)     // 4 waves x 136 floats (S[j] + 8 softmax rescale factors)
#define LDS_VT (LDS_SVEC + 544)       // 896 floats
#define LDS_RT (LDS_VT + VT_SIZE)     // rpe table 32 chunks x 8 x 4 = 1024 floats
#define LDS_TOTAL (LDS_RT + 1024)     // 39808 floats = 159232 bytes (<= 163840)

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
  const float *wl = wa + lane * 4;
  OPAQUE(wl);
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

// swizzled float offset of 16-byte chunk c (0..15) in a 256-byte staging row r (0..15)
__device__ __forceinline__ int sw_pos(int r, int c) { return (r * 16 + (c ^ (r & 15))) * 4; }

// MODE 0: layer 0, edge built in-kernel from the relative pose encoding (no edge read)
// MODE 1: layers 1..5, edge read from HBM
// update_mode 0: update every edge; 1: update only flagged columns; 2: run only flagged columns,
// no edge update (last layer).
#ifdef MIND_PAIR_TRACE
#define PT_DECL long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long pt_t = clock64();
#define PT(i) do { long long n_ = clock64(); pt_[i] += n_ - pt_t; pt_t = n_; } while (0)
#else
#define PT_DECL
#define PT(i)
#endif

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_pair(const PairJob *__restrict__ jobs, int n_jobs,
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
  for (int i = tid; i < 4096; i += 256) {
    ((f32x4 *)(lds + LDS_WAE))[i] = ((const f32x4 *)WAe)[i];
    if (update_mode != 2) ((f32x4 *)(lds + LDS_WAP))[i] = ((const f32x4 *)WAp)[i];
  }
  for (int i = tid; i < VT_SIZE; i += 256) lds[LDS_VT + i] = vtab[i];
  if (MODE == 0)
    for (int i = tid; i < 1024; i += 256) lds[LDS_RT + i] = rtab[i];
  __syncthreads();

  float *stage = lds + LDS_STAGE + wave * 1024;
  float *ptab = lds + LDS_PTAB + wave * 128;
  float *svec = lds + LDS_SVEC + wave * 136;
  const float *vt = lds + LDS_VT;

  PT_DECL
  for (int job = blockIdx.x * 4 + wave; job < n_jobs; job += gridDim.x * 4) {
    const PairJob J = jobs[job];
    if (update_mode == 2 && !(J.flags & 1)) continue;
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
        // ---- edge tile -> registers through the swizzled staging buffer: 2 x (16 rows x 256 B),
        //      every global access is a full 256-byte row segment (coalesced), LDS reads conflict-free
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const int r = 4 * n + (ll >> 4);
            const int cp = ll & 15;
            int irow = i0 + r;
            irow = irow < N ? irow : N - 1;
            const f32x4 v = *(const f32x4 *)(edge + (((size_t)J.edge_base + (size_t)irow * N + j) << 7) + hf * 64 + cp * 4);
            *(f32x4 *)(stage + sw_pos(r, cp)) = v;
          }
          LDS_FENCE();
#pragma unroll
          for (int b2 = 0; b2 < 4; ++b2) ef[4 * hf + b2] = *(const f32x4 *)(stage + sw_pos(lp, 4 * b2 + lq));
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
        // store through the staging buffer: full 256-byte row segments
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
          for (int b2 = 0; b2 < 4; ++b2) *(f32x4 *)(stage + sw_pos(lp, 4 * b2 + lq)) = up[4 * hf + b2];
          LDS_FENCE();
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const int r = 4 * n + (ll >> 4);
            const int cp = ll & 15;
            const f32x4 v = *(const f32x4 *)(stage + sw_pos(r, cp));
            if (i0 + r < N)
              *(f32x4 *)(edge + (((size_t)J.edge_base + (size_t)(i0 + r) * N + j) << 7) + hf * 64 + cp * 4) = v;
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
      //      transposing mem through the staging buffer one 64-feature half at a time
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) *(f32x4 *)(stage + sw_pos(lp, 4 * b2 + lq)) = mem[4 * hf + b2];
        LDS_FENCE();
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const float val = stage[sw_pos(t, ll >> 2) + (ll & 3)];
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
// 128 threads, thread t owns output feature t; weights are stored transposed [in][out] so that a
// wave reads 64 consecutive floats per k.
// =================================================================================================
#define TPW 8

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

__device__ __forceinline__ float block_sum128(float v, float *red, int tid) {
  // 128 threads = 2 waves
  v += __shfl_xor(v, 32, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1];
}

// y[t][tid] = sum_k WT[k*ldo + tid] * xin[t][k]  for t < TPW   (xin in LDS, row stride ldx)
template <int K>
__device__ __forceinline__ void matvec(float (&acc)[TPW], const float *__restrict__ WT, int ldo, int col,
                                       const float *xin, int ldx) {
#pragma unroll 16
  for (int k = 0; k < K; ++k) {
    const float w = WT[(size_t)k * ldo + col];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = fmaf(w, xin[t * ldx + k], acc[t]);
  }
}

// mode bits: 1 = init (x0 from actor/lane features), 2 = has epilogue, 4 = has prologue, 8 = only flagged
__global__ __launch_bounds__(128) void k_token(const TokMeta *__restrict__ meta, int n_tok, int mode,
                                               const float *__restrict__ actor_feat,
                                               const float *__restrict__ lane_feat, float *__restrict__ x,
                                               const float *__restrict__ part, float *__restrict__ ST,
                                               float *__restrict__ QK, TokWeights W) {
  __shared__ float xs[TPW][132];        // current token vectors
  __shared__ float tmp[TPW][260];       // scratch (o / h1 up to 256 wide)
  __shared__ float mb[TPW][8][132];     // normalised sum p*mem per head
  __shared__ float red[2];
  __shared__ float cw[TPW][8][8];       // combine weights per (token, head, split<=8)
  const int tid = threadIdx.x;
  const int tok0 = blockIdx.x * TPW;
  const int nt = min(TPW, n_tok - tok0);

  // ---- load x (or build x0)
  if (mode & 1) {
    // FusionNet projections (network.py:313-314, 323-324): Linear + LN + ReLU, cls token = zeros
    for (int t = 0; t < TPW; ++t) {
      float v = 0.f;
      if (t < nt) {
        const TokMeta m = meta[tok0 + t];
        if (m.type == 0) v = actor_feat[(size_t)m.src * 128 + tid];
        else if (m.type == 1) v = lane_feat[(size_t)m.src * 128 + tid];
      }
      tmp[t][tid] = v;
    }
    __syncthreads();
    float aa[TPW], al[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) { aa[t] = W.bpa[tid]; al[t] = W.bpl[tid]; }
    matvec<128>(aa, W.WpaT, 128, tid, &tmp[0][0], 260);
    matvec<128>(al, W.WplT, 128, tid, &tmp[0][0], 260);
    for (int t = 0; t < TPW; ++t) {
      int type = 2;
      if (t < nt) type = meta[tok0 + t].type;
      const float pre = type == 0 ? aa[t] : al[t];
      const float mean = block_sum128(pre, red, tid) * (1.0f / 128.0f);
      const float d = pre - mean;
      const float var = block_sum128(d * d, red, tid) * (1.0f / 128.0f);
      const float rstd = 1.0f / sqrtf(var + 1e-5f);
      const float g = type == 0 ? W.gpa[tid] : W.gpl[tid];
      const float b = type == 0 ? W.bepa[tid] : W.bepl[tid];
      float y = fmaxf(d * rstd * g + b, 0.f);
      if (type == 2) y = 0.f;
      xs[t][tid] = y;
    }
    __syncthreads();
  } else {
    for (int t = 0; t < TPW; ++t) xs[t][tid] = (t < nt) ? x[(size_t)(tok0 + t) * 128 + tid] : 0.f;
    __syncthreads();
  }

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
    for (int t = 0; t < TPW; ++t) {
      int ns = 0, slot0 = 0;
      if (t < nt) {
        const TokMeta m = meta[tok0 + t];
        ns = ((mode & 8) && !(m.flags & 1)) ? 0 : m.nsplit;
        slot0 = m.slot0;
      }
      for (int hd = 0; hd < 8; ++hd) {
        float v = 0.f;
        for (int s = 0; s < ns; ++s)
          v = fmaf(cw[t][hd][s], part[(size_t)(slot0 + s) * PART_STRIDE + 16 + hd * 128 + tid], v);
        mb[t][hd][tid] = v;
      }
    }
    __syncthreads();
    // ---- o = W_v,h mbar_h + b_v  (thread f = hd*16+d uses head hd = f>>4)
    {
      float acc[TPW];
      const float bvv = W.bv[tid];
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = bvv;
      const int hd = tid >> 4;
#pragma unroll 16
      for (int k = 0; k < 128; ++k) {
        const float w = W.WvT[k * 128 + tid];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = fmaf(w, mb[t][hd][k], acc[t]);
      }
#pragma unroll
      for (int t = 0; t < TPW; ++t) tmp[t][tid] = acc[t];
    }
    __syncthreads();
    // ---- att = W_o o + b_o ; x1 = LN2(x + att)
    {
      float acc[TPW];
      const float b = W.bo[tid];
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = b;
      matvec<128>(acc, W.WoT, 128, tid, &tmp[0][0], 260);
      const float g = W.g2[tid], be = W.b2[tid];
      for (int t = 0; t < TPW; ++t) {
        const float pre = xs[t][tid] + acc[t];
        const float mean = block_sum128(pre, red, tid) * (1.0f / 128.0f);
        const float d = pre - mean;
        const float var = block_sum128(d * d, red, tid) * (1.0f / 128.0f);
        xs[t][tid] = d * (1.0f / sqrtf(var + 1e-5f)) * g + be;
      }
      __syncthreads();
    }
    // ---- FFN 128 -> 256 -> 128, x2 = LN3(x1 + ff)
    {
      float a0[TPW], a1[TPW];
      const float b0 = W.b1[tid], b1v = W.b1[128 + tid];
#pragma unroll
      for (int t = 0; t < TPW; ++t) { a0[t] = b0; a1[t] = b1v; }
      matvec<128>(a0, W.W1T, 256, tid, &xs[0][0], 132);
      matvec<128>(a1, W.W1T, 256, 128 + tid, &xs[0][0], 132);
#pragma unroll
      for (int t = 0; t < TPW; ++t) { tmp[t][tid] = fmaxf(a0[t], 0.f); tmp[t][128 + tid] = fmaxf(a1[t], 0.f); }
      __syncthreads();
      float acc[TPW];
      const float b = W.bb2[tid];
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = b;
      matvec<256>(acc, W.W2T, 128, tid, &tmp[0][0], 260);
      const float g = W.g3[tid], be = W.b3[tid];
      for (int t = 0; t < TPW; ++t) {
        const float pre = xs[t][tid] + acc[t];
        const float mean = block_sum128(pre, red, tid) * (1.0f / 128.0f);
        const float d = pre - mean;
        const float var = block_sum128(d * d, red, tid) * (1.0f / 128.0f);
        xs[t][tid] = d * (1.0f / sqrtf(var + 1e-5f)) * g + be;
      }
      __syncthreads();
    }
  }
  // ---- write x
  for (int t = 0; t < nt; ++t) x[(size_t)(tok0 + t) * 128 + tid] = xs[t][tid];

  if (mode & 4) {
    // ---- prologue of the next layer: S = W_s x, T = W_t x + b_m, q = W_q x + b_q,
    //      qk[hd][f] = sum_d q[hd*16+d] W_k[hd*16+d][f] / 4      (scale 1/sqrt(16))
    float as[TPW], at[TPW], aq[TPW];
    const float bmv = W.bm[tid], bqv = W.bq[tid];
#pragma unroll
    for (int t = 0; t < TPW; ++t) { as[t] = 0.f; at[t] = bmv; aq[t] = bqv; }
    matvec<128>(as, W.WsT, 128, tid, &xs[0][0], 132);
    matvec<128>(at, W.WtT, 128, tid, &xs[0][0], 132);
    matvec<128>(aq, W.WqT, 128, tid, &xs[0][0], 132);
    for (int t = 0; t < nt; ++t) {
      ST[(size_t)(tok0 + t) * 256 + tid] = as[t];
      ST[(size_t)(tok0 + t) * 256 + 128 + tid] = at[t];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TPW; ++t) tmp[t][tid] = aq[t];
    __syncthreads();
    for (int hd = 0; hd < 8; ++hd) {
      float acc[TPW];
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = 0.f;
#pragma unroll
      for (int d = 0; d < 16; ++d) {
        const float w = W.Wk[(size_t)(hd * 16 + d) * 128 + tid];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = fmaf(w, tmp[t][hd * 16 + d], acc[t]);
      }
      for (int t = 0; t < nt; ++t) QK[(size_t)(tok0 + t) * 1024 + hd * 128 + tid] = acc[t] * 0.25f;
    }
  }
}
