// k_token_mfma: the per-token epilogue of fusion layer L (softmax merge of the column partials, V / out projections, LayerNorm,
// FFN 128 -> 256 -> 128, LayerNorm; network.py:177-179, 205-232) and the prologue of layer L+1 (S, T, q, folded K query) with every
// projection on the fp32 MFMA (v_mfma_f32_16x16x4_f32: fp32 products, fp32 accumulation -- the reference's arithmetic class).
//
// The VALU kernel (k_token, fusion_kernels.hip) re-reads every token vector once per output column group: its time is the LDS
// bandwidth of those broadcast reads (64 ds_read_b128 per thread and projection: ~1.7 us per projection and workgroup).  Here a
// workgroup owns 16 tokens, a projection is out^T [128 x 16] = W [128 x 128] x X^T [128 x 16]: the weights are the A operand in
// fragment order (one float4 load per lane = the A fragments of four MFMAs, packed on the host by pack_afrag), the token tile is the
// B operand (one ds_read_b128 per lane and 16 features), each of the four waves produces two 16-feature blocks of all 16 tokens.
// A token's result does not depend on the other tokens of its tile (every D element is its own dot product), so predictions stay
// independent of the batch composition.
//
// included by mind_hip.hip behind fusion_kernels.hip (TokMeta, TokWeights, PART_STRIDE, pk_bf16 ...)
#define TM_TOK 16
#define TM_THREADS 256
#define TM_LDX 132     // row stride of the 128-wide token tiles (floats)
#define TM_LDT 260     // row stride of the 256-wide scratch tile
#define TM_HP 8        // heads per pass of the partial-sum merge / V projection: the merged sums of TM_HP heads live in LDS at a time (all
                       // eight: 68 KB of the kernel's 97 KB = one four-wave workgroup per CU; two at a time = 46 KB = three workgroups
                       // per CU measured no faster on the cfg4 tree -- 234 vs 221 us per launch -- so one pass it is)

struct TokWeightsM {   // A-fragment packings (pack_afrag) of the matrices TokWeights holds transposed
  const float *Wv, *Wo, *W1a, *W1b, *W2a, *W2b;     // epilogue of layer L-1 (W1: output halves; W2: input halves)
  const float *Ws, *Wt, *Wq, *Wkf;                  // prologue of layer L (Wkf: per head [8][ob 8][lane 64][4], see pack_wk_frag)
  const float *Wpa, *Wpl;                           // init projections
};

typedef float tm_f4 __attribute__((ext_vector_type(4)));

// The A fragments (weights) of this wave's two output blocks of one projection: sixteen float4 per lane, requested together.  They do
// not depend on the activations, so a projection's fragments are requested BEFORE the barrier that ends the previous one.
struct TmFrag { tm_f4 a0[8], a1[8]; };
__device__ __forceinline__ void tm_load(TmFrag &F, const float *__restrict__ Wf, int ob0, int lane) {
  const tm_f4 *A = reinterpret_cast<const tm_f4 *>(Wf) + (size_t)ob0 * 8 * 64 + lane;
#pragma unroll
  for (int s4 = 0; s4 < 8; ++s4) { F.a0[s4] = A[s4 * 64]; F.a1[s4] = A[(8 + s4) * 64]; }
}
// acc[t] (+)= W[16 (ob0 + t) .. +16][0..128) x X^T, t = 0, 1.  xb: this lane's B row = token (lane & 15) of a [16][ldb] tile; a non-zero
// ob_stride selects a different tile per output block (the V projection reads head ob's normalised sum)
template <int BF>
__device__ __forceinline__ void tm_mma(tm_f4 (&acc)[2], const TmFrag &F, int ob0, const float *xb, int ob_stride, int lane) {
  if (BF) {
    // bf16 hi + lo split operands on v_mfma_f32_16x16x32_bf16 (the three significant products, fp32 accumulate: the arithmetic of
    // k_pair_bf): F.a?[g] = hi fragment of k-group g, F.a?[4 + g] = lo (pack_tok_bfrag); the B operand of a lane = 8 consecutive
    // features 32 g + 8 q .. of its token, split here
    const float *b0p = xb + 8 * (lane >> 4) + (size_t)ob0 * ob_stride;
    const float *b1p = b0p + ob_stride;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x4 bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 1 && !ob_stride) { bh[1] = bh[0]; bl[1] = bl[0]; break; }
        const float *bp = (t ? b1p : b0p) + 32 * g;
        const tm_f4 x0 = *reinterpret_cast<const tm_f4 *>(bp), x1 = *reinterpret_cast<const tm_f4 *>(bp + 4);
        const float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const u32 h = pk_bf16(v[2 * d], v[2 * d + 1]);
          bh[t][d] = h;
          bl[t][d] = pk_bf16(v[2 * d] - bf_lo_f32(h), v[2 * d + 1] - bf_hi_f32(h));
        }
      }
      const u32x4 a0h = __builtin_bit_cast(u32x4, F.a0[g]), a0l = __builtin_bit_cast(u32x4, F.a0[4 + g]);
      const u32x4 a1h = __builtin_bit_cast(u32x4, F.a1[g]), a1l = __builtin_bit_cast(u32x4, F.a1[4 + g]);
      acc[0] = MFMA_BF(a0l, bh[0], acc[0]);
      acc[1] = MFMA_BF(a1l, bh[1], acc[1]);
      acc[0] = MFMA_BF(a0h, bl[0], acc[0]);
      acc[1] = MFMA_BF(a1h, bl[1], acc[1]);
      acc[0] = MFMA_BF(a0h, bh[0], acc[0]);
      acc[1] = MFMA_BF(a1h, bh[1], acc[1]);
    }
    return;
  }
  const float *b0p = xb + 4 * (lane >> 4) + (size_t)ob0 * ob_stride;
  const float *b1p = b0p + ob_stride;
#pragma unroll
  for (int s4 = 0; s4 < 8; ++s4) {
    const tm_f4 b0 = *reinterpret_cast<const tm_f4 *>(b0p + 16 * s4);
    const tm_f4 b1 = ob_stride ? *reinterpret_cast<const tm_f4 *>(b1p + 16 * s4) : b0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a0[s4][w], b0[w], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a1[s4][w], b1[w], acc[1], 0, 0, 0);
    }
  }
}
template <int BF>
__device__ __forceinline__ void tm_gemm(tm_f4 (&acc)[2], const float *__restrict__ Wf, int ob0, const float *xb, int ob_stride, int lane) {
  TmFrag F;
  tm_load(F, Wf, ob0, lane);
  tm_mma<BF>(acc, F, ob0, xb, ob_stride, lane);
}

// D element i of a lane: feature 16 ob + 4 (lane >> 4) + i of token lane & 15
#define TM_FEAT(ob, i) (16 * (ob) + 4 * (lane >> 4) + (i))

// LayerNorm (eps 1e-5) over the 128 features of each of the 16 rows of `tile` [16][ld], in place; optional ReLU; rows whose type says
// so are zeroed (init: the cls token).  256 threads: 16 per row, 8 features each.  Two-pass (mean, then variance of the deviations).
__device__ __forceinline__ void tm_layernorm(float *tile, int ld, const float *__restrict__ g, const float *__restrict__ be, bool relu, int tid) {
  const int row = tid >> 4, part = tid & 15;
  float *p = tile + row * ld + part * 8;
  float v[8], s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) { v[k] = p[k]; s += v[k]; }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s * (1.0f / 128.0f);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) { v[k] -= mean; q += v[k] * v[k]; }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float inv = 1.0f / sqrtf(q * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float y = v[k] * inv * g[part * 8 + k] + be[part * 8 + k];
    p[k] = relu ? fmaxf(y, 0.f) : y;
  }
}

// BF = 0: fp32 MFMA (opt-in, mind_set_tuning("tok_mfma")); BF = 1: bf16 hi + lo split operands, the token kernel of scenes of at least
// tok_bf_min_n tokens under the bf16 pair-kernel arithmetics (the folded K query stays on the fp32 MFMA: 16-wide contractions)
template <int BF>
__global__ __launch_bounds__(TM_THREADS) void k_token_mfma(const TokMeta *__restrict__ meta, int n_tok, int mode, const float *__restrict__ actor_feat,
                                                           const float *__restrict__ lane_feat, float *__restrict__ x, const float *__restrict__ part,
                                                           float *__restrict__ ST, float *__restrict__ QK, TokWeights W, TokWeightsM WM) {
  extern __shared__ __attribute__((aligned(16))) float tm_sm[];
  float *xs = tm_sm;                              // [16][132]  current token vectors
  float *tmp = xs + TM_TOK * TM_LDX;              // [16][260]  scratch (o / h1 / q)
  float *mb = tmp + TM_TOK * TM_LDT;              // [16][TM_HP][132] normalised sum p*mem per head, TM_HP heads per pass
  float *cw = mb + TM_TOK * TM_HP * TM_LDX;       // [16][8][8]  combine weights per (token, head, split)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tok0 = blockIdx.x * TM_TOK;
  const int nt = min(TM_TOK, n_tok - tok0);
  const int ob0 = 2 * wave;                       // this wave's two 16-feature output blocks
  const int tk = lane & 15;                       // the token (tile row) of this lane's D elements / B operand
  const bool tk_ok = tk < nt;
  TmFrag Fv, Fo;                 // weight fragments in flight across the phase boundaries (see tm_load)
  bool pre = false;              // Fv / Fo already hold the prologue's W_s / W_t

  // ---- load x (or build x0)
  if (mode & 1) {
    // FusionNet projections (network.py:313-314, 323-324): Linear + LN + ReLU per token type, cls token = zeros
    for (int e = tid; e < TM_TOK * 128; e += TM_THREADS) {
      const int t = e >> 7, col = e & 127;
      float f = 0.f;
      if (t < nt) {
        const TokMeta m = meta[tok0 + t];
        if (m.type == 0) f = actor_feat[(size_t)m.src * 128 + col];
        else if (m.type == 1) f = lane_feat[(size_t)m.src * 128 + col];
      }
      tmp[t * TM_LDT + col] = f;
    }
    __syncthreads();
    const int ty = tk_ok ? meta[tok0 + tk].type : 2;
    tm_f4 aa[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, al[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    tm_gemm<BF>(aa, WM.Wpa, ob0, tmp + tk * TM_LDT, 0, lane);
    tm_gemm<BF>(al, WM.Wpl, ob0, tmp + tk * TM_LDT, 0, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = TM_FEAT(ob0 + t, i);
        xs[tk * TM_LDX + f] = ty == 0 ? aa[t][i] + W.bpa[f] : al[t][i] + W.bpl[f];
      }
    __syncthreads();
    {
      // LN with per-token (type-dependent) affine
      const int row = tid >> 4, prt = tid & 15;
      const int tyr = row < nt ? meta[tok0 + row].type : 2;
      tm_layernorm(xs, TM_LDX, tyr == 0 ? W.gpa : W.gpl, tyr == 0 ? W.bepa : W.bepl, true, tid);
      if (tyr == 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) xs[row * TM_LDX + prt * 8 + k] = 0.f;
      }
    }
    __syncthreads();
  } else {
    for (int e = tid; e < TM_TOK * 128; e += TM_THREADS) {
      const int t = e >> 7, col = e & 127;
      xs[t * TM_LDX + col] = t < nt ? x[(size_t)(tok0 + t) * 128 + col] : 0.f;
    }
    __syncthreads();
  }
  if (mode & 2) {
    // ---- combine split partials: weights exp(m_s - M) / L  (softmax over i finished here)
    if (tid < TM_TOK * 8) {
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
      for (int s = 0; s < 8; ++s) cw[(t * 8 + hd) * 8 + s] = wgt[s];
    }
    __syncthreads();
    tm_load(Fv, WM.Wv, ob0, lane);       // the V projection's weights travel while the partials are combined
    TmFrag F1a, F1b;
    // TM_HP heads per pass: merge their partial sums into mb, then the waves whose output blocks (= heads) they are run the V projection
    // o = W_v,h mbar_h + b_v  (output block ob = head ob reads that head's normalised sum)
    for (int h0_ = 0; h0_ < 8; h0_ += TM_HP) {
      {
        // mb[t][hd - h0_][col] = sum_s cw[t][hd][s] part[slot0 + s][16 + hd * 128 + col]: thread = (token, group of 8 columns); the
        // float4 pairs of a split's TM_HP heads are requested together
        const int t = tid >> 4, c8 = (tid & 15) * 8;
        int ns = 0, slot0 = 0;
        if (t < nt) {
          const TokMeta m = meta[tok0 + t];
          ns = ((mode & 8) && !(m.flags & 1)) ? 0 : m.nsplit;
          slot0 = m.slot0;
        }
        tm_f4 v[TM_HP][2];
#pragma unroll
        for (int hd = 0; hd < TM_HP; ++hd) { v[hd][0] = tm_f4{0, 0, 0, 0}; v[hd][1] = tm_f4{0, 0, 0, 0}; }
        for (int sp = 0; sp < ns; ++sp) {
          const tm_f4 *ps = reinterpret_cast<const tm_f4 *>(part + (size_t)(slot0 + sp) * PART_STRIDE + 16 + (size_t)h0_ * 128 + c8);
          tm_f4 pv[TM_HP][2];
#pragma unroll
          for (int hd = 0; hd < TM_HP; ++hd) { pv[hd][0] = ps[hd * 32]; pv[hd][1] = ps[hd * 32 + 1]; }
#pragma unroll
          for (int hd = 0; hd < TM_HP; ++hd) {
            const float wgt = cw[(t * 8 + h0_ + hd) * 8 + sp];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[hd][0][k] = fmaf(wgt, pv[hd][0][k], v[hd][0][k]); v[hd][1][k] = fmaf(wgt, pv[hd][1][k], v[hd][1][k]); }
          }
        }
#pragma unroll
        for (int hd = 0; hd < TM_HP; ++hd) {
          *reinterpret_cast<tm_f4 *>(mb + (t * TM_HP + hd) * TM_LDX + c8) = v[hd][0];
          *reinterpret_cast<tm_f4 *>(mb + (t * TM_HP + hd) * TM_LDX + c8 + 4) = v[hd][1];
        }
      }
      __syncthreads();
      if (ob0 >= h0_ && ob0 < h0_ + TM_HP) {       // (wave-uniform: this wave's two output blocks are heads ob0, ob0 + 1)
        tm_f4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        tm_mma<BF>(acc, Fv, ob0 - h0_, mb + (size_t)tk * TM_HP * TM_LDX, TM_LDX, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) { const int f = TM_FEAT(ob0 + t, i); tmp[tk * TM_LDT + f] = acc[t][i] + W.bv[f]; }
      }
      __syncthreads();
    }
    tm_load(Fo, WM.Wo, ob0, lane);
    // ---- att = W_o o + b_o ; x1 = LN2(x + att)
    {
      tm_f4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      tm_mma<BF>(acc, Fo, ob0, tmp + tk * TM_LDT, 0, lane);
      tm_load(F1a, WM.W1a, ob0, lane);
      tm_load(F1b, WM.W1b, ob0, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int f = TM_FEAT(ob0 + t, i); xs[tk * TM_LDX + f] += acc[t][i] + W.bo[f]; }
    }
    __syncthreads();
    tm_layernorm(xs, TM_LDX, W.g2, W.b2, false, tid);
    __syncthreads();
    // ---- FFN 128 -> 256 -> 128, x2 = LN3(x1 + ff)
    {
      tm_f4 h0[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, h1[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      tm_mma<BF>(h0, F1a, ob0, xs + tk * TM_LDX, 0, lane);
      tm_load(F1a, WM.W2a, ob0, lane);                  // (the fragment registers of W1a now carry W2a, those of W1b W2b)
      tm_mma<BF>(h1, F1b, ob0, xs + tk * TM_LDX, 0, lane);
      tm_load(F1b, WM.W2b, ob0, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = TM_FEAT(ob0 + t, i);
          tmp[tk * TM_LDT + f] = fmaxf(h0[t][i] + W.b1[f], 0.f);
          tmp[tk * TM_LDT + 128 + f] = fmaxf(h1[t][i] + W.b1[128 + f], 0.f);
        }
    }
    __syncthreads();
    {
      tm_f4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      tm_mma<BF>(acc, F1a, ob0, tmp + tk * TM_LDT, 0, lane);
      tm_mma<BF>(acc, F1b, ob0, tmp + tk * TM_LDT + 128, 0, lane);
      if (mode & 4) { tm_load(Fv, WM.Ws, ob0, lane); tm_load(Fo, WM.Wt, ob0, lane); }      // the prologue's first two projections
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int f = TM_FEAT(ob0 + t, i); xs[tk * TM_LDX + f] += acc[t][i] + W.bb2[f]; }
    }
    __syncthreads();
    tm_layernorm(xs, TM_LDX, W.g3, W.b3, false, tid);
    __syncthreads();
    pre = true;
  }
  // ---- write x
  for (int e = tid; e < nt * 128; e += TM_THREADS) x[(size_t)tok0 * 128 + e] = xs[(e >> 7) * TM_LDX + (e & 127)];

  if (mode & 4) {
    // ---- prologue of the next layer: S = W_s x, T = W_t x + b_m, q = W_q x + b_q, qk[hd][f] = sum_d q[hd*16+d] W_k[hd*16+d][f] / 4
    {
      tm_f4 as[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, at[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, aq[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      TmFrag Fq;
      tm_load(Fq, WM.Wq, ob0, lane);
      if (!pre) { tm_load(Fv, WM.Ws, ob0, lane); tm_load(Fo, WM.Wt, ob0, lane); }
      tm_mma<BF>(as, Fv, ob0, xs + tk * TM_LDX, 0, lane);
      tm_mma<BF>(at, Fo, ob0, xs + tk * TM_LDX, 0, lane);
      tm_mma<BF>(aq, Fq, ob0, xs + tk * TM_LDX, 0, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = TM_FEAT(ob0 + t, i);
          if (tk_ok) {
            ST[(size_t)(tok0 + tk) * 256 + f] = as[t][i];
            ST[(size_t)(tok0 + tk) * 256 + 128 + f] = at[t][i] + W.bm[f];
          }
          tmp[tk * TM_LDT + f] = aq[t][i] + W.bq[f];
        }
    }
    __syncthreads();
    // folded K query: per head a [128 features] x [16 d] x [16 tokens] product (4 MFMAs per output block); the eight heads in turn
    const tm_f4 *Wk4 = reinterpret_cast<const tm_f4 *>(WM.Wkf);
    for (int hd = 0; hd < 8; ++hd) {
      const tm_f4 b = *reinterpret_cast<const tm_f4 *>(tmp + tk * TM_LDT + hd * 16 + 4 * (lane >> 4));
      tm_f4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      const tm_f4 a0 = Wk4[((size_t)hd * 8 + ob0) * 64 + lane], a1 = Wk4[((size_t)hd * 8 + ob0 + 1) * 64 + lane];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[w], b[w], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[w], b[w], acc[1], 0, 0, 0);
      }
      if (!tk_ok) continue;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int col = TM_FEAT(ob0 + t, i);
          const float v = acc[t][i] * 0.25f;
          if (mode & 16) {
            // bf16 hi / lo parts in the A-operand order of k_pair_bf (pair_bf16_kernels.hip): per token 2048 x u16 =
            // [part 2][k-group 4][row 32 = head 8 x lane quarter 4][slot 8], feature 16 (2g + (i >> 2)) + 4q + (i & 3) <-> slot i
            const int g = col >> 5, qq = (col >> 2) & 3, ii = 4 * ((col >> 4) & 1) + (col & 3);
            const int idx = ((g * 32) + hd * 4 + qq) * 8 + ii;
            const u32 h = pk_bf16(v, v) & 0xffffu;
            const float r1 = v - bf_lo_f32(h);
            const u32 l = pk_bf16(r1, 0.f) & 0xffffu;
            unsigned short *qs = reinterpret_cast<unsigned short *>(QK + (size_t)(tok0 + tk) * ((mode & 32) ? 1536 : 1024));
            qs[idx] = (unsigned short)h;
            qs[1024 + idx] = (unsigned short)l;
            if (mode & 32) qs[2048 + idx] = (unsigned short)(pk_bf16(r1 - bf_lo_f32(l), 0.f) & 0xffffu);      // third part (k_pair_t6)
          } else {
            QK[(size_t)(tok0 + tk) * 1024 + hd * 128 + col] = v;
          }
        }
    }
  }
}

static inline size_t mind_token_mfma_lds_bytes() {
  return (size_t)(TM_TOK * TM_LDX + TM_TOK * TM_LDT + TM_TOK * TM_HP * TM_LDX + TM_TOK * 64) * sizeof(float);
}
