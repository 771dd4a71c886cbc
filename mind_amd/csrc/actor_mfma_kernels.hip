// ActorNet (Res1d groups + FPN + output Res1d) on the bf16 MFMA, fp32-accurate by splitting both operands into three bf16 parts
// (x = hi + mid + lo exactly; six partial products per term, fp32 accumulate), one workgroup of 16 waves per actor.
//
// Reference semantics: planners/mind/networks/network.py:20-61 (ActorNet), layers.py:127-188 (Conv1d / Res1d with
// GroupNorm(1 group)); same math as k_actor_net (encdec_kernels.hip), which stays the fp32 (VALU) arithmetic.
//
// A Conv1d is a GEMM  out[co][t] = sum_k Wm[co][k] X[k][t],  k = dk * Cin_pad + ci,  X[k][t] = in[ci][t*stride + dk - pad]:
//   * M = output channels: the weights are the MFMA A operand, packed on the host into fragment order
//     [m-tile][k-step][part hi/mid/lo][lane 64][4 dwords] (pack_conv_frag, mind_hip.hip): one 16-byte load per lane per
//     fragment, coalesced 1 KB per wave; lane (r = lane & 15, q = lane >> 4) holds row co = 16 mt + r, k-slots 8 q .. 8 q + 7.
//   * N = time: the activations are the B operand.  Every activation tensor lives in LDS as a time-major image of three bf16
//     planes, row t = [hi: C x bf16][mid: C x bf16][lo: C x bf16][16 bytes pad]: the split is done ONCE by the layer that
//     produces the tensor (not by each of the Cout/16 tiles that consume it), and the 8 k-slots of a lane -- 8 consecutive input
//     channels at one tap -- are one ds_read_b128 per plane; rows outside [0, Tin) (the conv padding) and k-slots past
//     ksz * Cin_pad are zeros.
//   * C/D: lane holds time column t = 16 nt + (lane & 15), channels 16 mt + 4 (lane >> 4) + (0..3): the GroupNorm statistics are
//     reduced straight from the accumulators (two block reductions), the normalised tile (+ residual, + the FPN's upsampled
//     upper level, ReLU) is split and written once, 8 bytes per lane and plane.
// LDS per workgroup: 110.5 KB (AM_* below).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AM_T 1024
#ifdef MIND_ACTOR_TRACE
__device__ long long am_tr_[64];
__device__ int am_trn_;
#define AM_MARK() do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long n_ = clock64(); am_tr_[am_trn_ & 63] = n_; ++am_trn_; } } while (0)
#else
#define AM_MARK() do {} while (0)
#endif
#define AM_WAVES 16
#define AM_MAXT 2                           // output tiles per wave (24 tiles of the 128 x 48 layers / 16 waves)
#define AM_RSD(C) (3 * (C) / 2 + 4)         // row length of a split image in dwords
// LDS carve (dwords)
#define AM_BUF 2496                         // one split image of the Res1d groups: 48 x AM_RSD(32) is the largest
#define AM_XIN 0                            // [48] x AM_RSD(16)   (14 input channels padded to 16)
#define AM_O0 1344                          // o0 o1 o2 o3 ta tb tc
#define AM_FA (AM_O0 + 7 * AM_BUF)          // [48] x AM_RSD(128): FPN level 0
#define AM_RED (AM_FA + 48 * AM_RSD(128))
#define AM_LDS_DWORDS (AM_RED + 64)

struct AmRes { const u32 *c1, *c2, *ds; const float *g1, *b1, *g2, *b2, *gd, *bd; };
struct AmLat { const u32 *w; const float *g, *b; };
struct AmW {                                // packed conv fragments + GroupNorm affine, same order as ActorW
  AmRes res[9];
  AmLat lat[4];
};
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float am_block_sum(float v, float *red, int flip) {
  v = wave_sum(v);
  float *r = red + (flip & 1) * 32;
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < AM_WAVES; ++w) s += r[w];
  return s;
}

// y (4 consecutive channels of one time column) -> the three bf16 planes of row `rowp` (dword pointer to the row's channel pair)
template <int NP>
__device__ __forceinline__ void am_store_split(u32 *rowp, int plane, const f32x4 &y) {
  u32x2 h, m, l;
  h[0] = pk_bf16(y[0], y[1]); h[1] = pk_bf16(y[2], y[3]);
  *(u32x2 *)rowp = h;
  if (NP == 1) return;
  const float r0 = y[0] - bf_lo_f32(h[0]), r1 = y[1] - bf_hi_f32(h[0]), r2 = y[2] - bf_lo_f32(h[1]), r3 = y[3] - bf_hi_f32(h[1]);
  m[0] = pk_bf16(r0, r1); m[1] = pk_bf16(r2, r3);
  *(u32x2 *)(rowp + plane) = m;
  l[0] = pk_bf16(r0 - bf_lo_f32(m[0]), r1 - bf_hi_f32(m[0])); l[1] = pk_bf16(r2 - bf_lo_f32(m[1]), r3 - bf_hi_f32(m[1]));
  *(u32x2 *)(rowp + 2 * plane) = l;
}
// the fp32 values back (hi + mid + lo is exact)
template <int NP>
__device__ __forceinline__ f32x4 am_load_split(const u32 *rowp, int plane) {
  const u32x2 h = *(const u32x2 *)rowp;
  f32x4 y = {bf_lo_f32(h[0]), bf_hi_f32(h[0]), bf_lo_f32(h[1]), bf_hi_f32(h[1])};
  if (NP == 1) return y;
  const u32x2 m = *(const u32x2 *)(rowp + plane), l = *(const u32x2 *)(rowp + 2 * plane);
  y[0] = (y[0] + bf_lo_f32(m[0])) + bf_lo_f32(l[0]); y[1] = (y[1] + bf_hi_f32(m[0])) + bf_hi_f32(l[0]);
  y[2] = (y[2] + bf_lo_f32(m[1])) + bf_lo_f32(l[1]); y[3] = (y[3] + bf_hi_f32(m[1])) + bf_hi_f32(l[1]);
  return y;
}

// raw conv tiles of this wave: acc[i] = tile (wave + 16 i) of the (Cout/16) x ceil(Tout/16) grid, time tile fastest.
// NP = number of partial products per term: 6 = both operands split three ways (hi.hi + hi.mid + mid.hi + mid.mid + hi.lo +
// lo.hi: ~2^-24 relative, fp32-class), 3 = two-way split (hi.hi + hi.mid + mid.hi: ~2^-16), 1 = plain bf16 operands.  The
// leading products and the corrections run in two accumulators (two independent MFMA chains), summed at the end.
// Measured (tools/gpu.sh trace actor, profiles/r03ad): the convolutions are bound by this weight-fragment stream -- 3 KB per tile and
// k-step (hi / mid / lo), 9.1 MB per actor through ONE CU's 64 B/clk L2 port = 59 us of the kernel's 140, the GroupNorm stages
// (two block reductions + three barriers each, 26 of them) are another 55 us.  Requesting the fragments four k-steps ahead in a
// register ring changed nothing (144 vs 142 us): it is the port's bandwidth, not the round trips.
template <int NP, int LGC /*log2 Cin_pad*/, int KSZ, int STRIDE>
__device__ __forceinline__ void am_conv(const u32 *in, int Tin, const u32 *__restrict__ Wf, int Cout, int Tout,
                                        f32x4 (&acc)[AM_MAXT]) {
  constexpr int CP = 1 << LGC, RSD = AM_RSD(CP), PLANE = CP / 2, PAD = (KSZ - 1) / 2;
  constexpr int KS = (KSZ * CP + 31) / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int ntt = (Tout + 15) >> 4;
  const int tiles = (Cout >> 4) * ntt;
#pragma unroll
  for (int i = 0; i < AM_MAXT; ++i) {
    acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ti = wave + AM_WAVES * i;
    if (ti >= tiles) continue;
    const int mt = ti / ntt, nt = ti - mt * ntt;
    const int t = nt * 16 + r;
    const u32 *wp = Wf + (size_t)mt * KS * 768 + lane * 4;
    f32x4 corr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 ah = *(const u32x4 *)(wp + (size_t)ks * 768);
      u32x4 am, al;
      if (NP >= 3) am = *(const u32x4 *)(wp + (size_t)ks * 768 + 256);
      if (NP == 6) al = *(const u32x4 *)(wp + (size_t)ks * 768 + 512);
      const int k0 = ks * 32 + q * 8;
      const int dk = k0 >> LGC, ci = k0 & (CP - 1);
      const int row = t * STRIDE + dk - PAD;
      u32x4 bh = {0u, 0u, 0u, 0u}, bm = {0u, 0u, 0u, 0u}, bl = {0u, 0u, 0u, 0u};
      if (dk < KSZ && t < Tout && row >= 0 && row < Tin) {
        const u32 *bp = in + row * RSD + (ci >> 1);
        bh = *(const u32x4 *)bp;
        if (NP >= 3) bm = *(const u32x4 *)(bp + PLANE);
        if (NP == 6) bl = *(const u32x4 *)(bp + 2 * PLANE);
      }
      acc[i] = MFMA_BF(ah, bh, acc[i]);
      if (NP >= 3) {
        if (NP == 6) {
          corr = MFMA_BF(al, bh, corr);
          corr = MFMA_BF(ah, bl, corr);
          corr = MFMA_BF(am, bm, corr);
        }
        corr = MFMA_BF(am, bh, corr);
        corr = MFMA_BF(ah, bm, corr);
      }
    }
    if (NP >= 3) acc[i] += corr;
  }
}

// GroupNorm(1 group) over the Cout x Tout outputs held in the accumulators, per-channel affine, optional residual (a split image
// with the same row length), optional x2 linear upsample (align_corners = False) of `up` ([Tout/2][132] fp32) added, optional
// ReLU.  Output: `outs` (split image [Tout] x AM_RSD(Cout)), or `outf` (fp32 [Tout][132], FPN levels that no convolution reads),
// or `gout` (final layer: only the last time column, to global memory).  Ends with a barrier.
template <int NP>
__device__ __forceinline__ void am_gn(f32x4 (&acc)[AM_MAXT], int Cout, int Tout, const float *__restrict__ g,
                                      const float *__restrict__ b, const u32 *resid, const float *up, bool relu, u32 *outs,
                                      float *outf, float *__restrict__ gout, float *red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int ntt = (Tout + 15) >> 4;
  const int tiles = (Cout >> 4) * ntt;
  const int RSD = AM_RSD(Cout), PLANE = Cout >> 1;
  f32x4 gg[AM_MAXT], bb[AM_MAXT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < AM_MAXT; ++i) {
    const int ti = wave + AM_WAVES * i;
    if (ti >= tiles) continue;
    const int mt = ti / ntt, nt = ti - mt * ntt;
    gg[i] = *(const f32x4 *)(g + mt * 16 + q * 4);
    bb[i] = *(const f32x4 *)(b + mt * 16 + q * 4);
    if (nt * 16 + r < Tout) s += (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
  }
  const float n = (float)(Cout * Tout);
  const float mean = am_block_sum(s, red, 0) / n;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < AM_MAXT; ++i) {
    const int ti = wave + AM_WAVES * i;
    if (ti >= tiles) continue;
    const int nt = ti % ntt;
    if (nt * 16 + r < Tout) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = acc[i][e] - mean; v = fmaf(d, d, v); }
    }
  }
  const float rstd = 1.0f / sqrtf(am_block_sum(v, red, 1) / n + 1e-5f);
#pragma unroll
  for (int i = 0; i < AM_MAXT; ++i) {
    const int ti = wave + AM_WAVES * i;
    if (ti >= tiles) continue;
    const int mt = ti / ntt, nt = ti - mt * ntt;
    const int t = nt * 16 + r, co = mt * 16 + q * 4;
    if (t >= Tout) continue;
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = (acc[i][e] - mean) * rstd * gg[i][e] + bb[i][e];
    if (resid) y += am_load_split<NP>(resid + t * RSD + (co >> 1), PLANE);
    if (up) {
      const int Th = Tout >> 1;
      float sp = (t + 0.5f) * 0.5f - 0.5f;
      sp = sp < 0.f ? 0.f : sp;
      const int i0 = (int)sp;
      const int i1 = i0 + 1 < Th ? i0 + 1 : Th - 1;
      const float l1 = sp - (float)i0;
      const f32x4 a0 = *(const f32x4 *)(up + i0 * 132 + co), a1 = *(const f32x4 *)(up + i1 * 132 + co);
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] += (1.0f - l1) * a0[e] + l1 * a1[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
    }
    if (gout) {
      if (t == Tout - 1) *(f32x4 *)(gout + co) = y;
    } else if (outf) {
      *(f32x4 *)(outf + t * 132 + co) = y;
    } else {
      am_store_split<NP>(outs + t * RSD + (co >> 1), PLANE, y);
    }
  }
  __syncthreads();
}

// Res1d (layers.py:175-188): in [Tin] x AM_RSD(Cin_pad) -> out [Tout] x AM_RSD(Cout); scratch t1, t2
template <int NP, int LGCI, int LGCO, int STRIDE, bool DS>
__device__ __forceinline__ void am_res(const u32 *in, int Tin, const AmRes &W, u32 *out, u32 *t1, u32 *t2, float *red) {
  constexpr int Cout = 1 << LGCO;
  const int Tout = Tin / STRIDE;
  f32x4 acc[AM_MAXT];
  am_conv<NP, LGCI, 3, STRIDE>(in, Tin, W.c1, Cout, Tout, acc);
  AM_MARK();
  am_gn<NP>(acc, Cout, Tout, W.g1, W.b1, nullptr, nullptr, true, t1, nullptr, nullptr, red);
  AM_MARK();
  const u32 *resid = in;
  if (DS) {
    am_conv<NP, LGCI, 1, STRIDE>(in, Tin, W.ds, Cout, Tout, acc);
    AM_MARK();
    am_gn<NP>(acc, Cout, Tout, W.gd, W.bd, nullptr, nullptr, false, t2, nullptr, nullptr, red);
    AM_MARK();
    resid = t2;
  }
  am_conv<NP, LGCO, 3, 1>(t1, Tout, W.c2, Cout, Tout, acc);
  AM_MARK();
  am_gn<NP>(acc, Cout, Tout, W.g2, W.b2, resid, nullptr, true, out, nullptr, nullptr, red);
  AM_MARK();
}

// FPN level (network.py:55-58): lateral conv + GroupNorm of `src` ([T] x AM_RSD(C)) plus the upsampled level above (`up`, fp32
// [T/2][132]; null at the top) -> `dstf` (fp32 [T][132]) or, for level 0 which the output Res1d convolves, the split image `dsts`
template <int NP, int LGC>
__device__ __forceinline__ void am_lateral(const u32 *src, int T, const AmLat &W, const float *up, float *dstf, u32 *dsts,
                                           float *red) {
  f32x4 acc[AM_MAXT];
  am_conv<NP, LGC, 3, 1>(src, T, W.w, 128, T, acc);
  AM_MARK();
  am_gn<NP>(acc, 128, T, W.g, W.b, nullptr, up, false, dsts, dstf, nullptr, red);
  AM_MARK();
}

template <int NP>
__global__ __launch_bounds__(AM_T) void k_actor_mfma(const float *__restrict__ actors /*[A,14,48]*/, int n_actors,
                                                     float *__restrict__ out /*[A,128]*/, AmW W) {
  extern __shared__ __attribute__((aligned(16))) u32 smu[];
  u32 *xin = smu + AM_XIN;
  u32 *o0 = smu + AM_O0, *o1 = o0 + AM_BUF, *o2 = o1 + AM_BUF, *o3 = o2 + AM_BUF;
  u32 *ta = o3 + AM_BUF, *tb = ta + AM_BUF, *tc = tb + AM_BUF;
  u32 *fa = smu + AM_FA;
  float *red = (float *)(smu + AM_RED);
  const int tid = threadIdx.x;
  const int a = blockIdx.x;
  if (a >= n_actors) return;
  // input [14][48] -> split image [48] x AM_RSD(16), channels 14, 15 zero: a thread takes two channels of one time step
  if (tid < 48 * 8) {
    const int t = tid >> 3, c = (tid & 7) * 2;
    const float v0 = c < 14 ? actors[((size_t)a * 14 + c) * 48 + t] : 0.f;
    const float v1 = c + 1 < 14 ? actors[((size_t)a * 14 + c + 1) * 48 + t] : 0.f;
    u32 *p = xin + t * AM_RSD(16) + (c >> 1);
    const u32 h = pk_bf16(v0, v1);
    const float r0 = v0 - bf_lo_f32(h), r1 = v1 - bf_hi_f32(h);
    const u32 m = pk_bf16(r0, r1);
    p[0] = h; p[8] = m; p[16] = pk_bf16(r0 - bf_lo_f32(m), r1 - bf_hi_f32(m));
  }
  __syncthreads();
#ifdef MIND_ACTOR_TRACE
  if (blockIdx.x == 0 && tid == 0) am_trn_ = 0;
#endif
  AM_MARK();
  am_res<NP, 4, 5, 1, true>(xin, 48, W.res[0], ta, tb, tc, red);
  am_res<NP, 5, 5, 1, false>(ta, 48, W.res[1], o0, tb, tc, red);
  am_res<NP, 5, 6, 2, true>(o0, 48, W.res[2], ta, tb, tc, red);
  am_res<NP, 6, 6, 1, false>(ta, 24, W.res[3], o1, tb, tc, red);
  am_res<NP, 6, 7, 2, true>(o1, 24, W.res[4], ta, tb, tc, red);
  am_res<NP, 7, 7, 1, false>(ta, 12, W.res[5], o2, tb, tc, red);
  am_res<NP, 7, 8, 2, true>(o2, 12, W.res[6], ta, tb, tc, red);
  am_res<NP, 8, 8, 1, false>(ta, 6, W.res[7], o3, tb, tc, red);
  // FPN top-down, fp32 levels 3 -> ta, 2 -> tb, 1 -> o2..o3 (dead by then); level 0 -> fa as a split image
  am_lateral<NP, 8>(o3, 6, W.lat[3], nullptr, (float *)ta, nullptr, red);
  am_lateral<NP, 7>(o2, 12, W.lat[2], (const float *)ta, (float *)tb, nullptr, red);
  am_lateral<NP, 6>(o1, 24, W.lat[1], (const float *)tb, (float *)o2, nullptr, red);
  am_lateral<NP, 5>(o0, 48, W.lat[0], (const float *)o2, nullptr, fa, red);
  // output Res1d(128, 128) at T = 48 (network.py:60); only the last time column is kept.  conv1's output goes to o0..o3 (dead)
  {
    const AmRes &R = W.res[8];
    f32x4 acc[AM_MAXT];
    am_conv<NP, 7, 3, 1>(fa, 48, R.c1, 128, 48, acc);
    AM_MARK();
    am_gn<NP>(acc, 128, 48, R.g1, R.b1, nullptr, nullptr, true, o0, nullptr, nullptr, red);
    AM_MARK();
    am_conv<NP, 7, 3, 1>(o0, 48, R.c2, 128, 48, acc);
    AM_MARK();
    am_gn<NP>(acc, 128, 48, R.g2, R.b2, fa, nullptr, true, nullptr, nullptr, out + (size_t)a * 128, red);
    AM_MARK();
  }
#ifdef MIND_ACTOR_TRACE
  if (blockIdx.x == 0 && tid == 0) {
    printf("[k_actor_mfma<%d>] cycles conv/gn per stage:", NP);
    for (int q = 1; q < am_trn_ && q < 64; ++q) printf(" %lld", am_tr_[q] - am_tr_[q - 1]);
    printf("\n");
  }
#endif
}
extern "C" size_t mind_actor_mfma_lds_bytes() { return (size_t)AM_LDS_DWORDS * sizeof(u32); }
