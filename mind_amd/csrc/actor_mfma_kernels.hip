// ActorNet (Res1d groups + FPN + output Res1d) on the bf16 MFMA, fp32-accurate by splitting both operands into three bf16 parts
// (x = hi + mid + lo exactly; six partial products per term, fp32 accumulate), one workgroup of 8 waves per actor, two
// workgroups per CU.
//
// Reference semantics: planners/mind/networks/network.py:20-61 (ActorNet), layers.py:127-188 (Conv1d / Res1d with
// GroupNorm(1 group)); same math as k_actor_net (encdec_kernels.hip), which stays the fp32 (VALU) arithmetic.
//
// A Conv1d is a GEMM  out[co][t] = sum_k Wm[co][k] X[k][t],  k = dk * Cin_pad + ci,  X[k][t] = in[ci][t*stride + dk - pad]:
//   * M = output channels: the weights are the MFMA A operand, packed on the host into fragment order
//     [m-tile][k-step][part hi/mid/lo][lane 64][4 dwords] (pack_conv_frag, mind_hip.hip): one 16-byte load per lane per
//     fragment, coalesced 1 KB per wave; lane (r = lane & 15, q = lane >> 4) holds row co = 16 mt + r, k-slots 8 q .. 8 q + 7.
//   * N = time: the activations are the B operand, read from a time-major LDS image [t][C + 4]: the 8 k-slots of a lane are 8
//     consecutive input channels at one tap, i.e. two ds_read_b128; rows outside [0, Tin) (the conv padding) and k-slots past
//     ksz * Cin_pad are zeros.
//   * C/D: lane holds time column t = 16 nt + (lane & 15), channels 16 mt + 4 (lane >> 4) + (0..3): the GroupNorm statistics are
//     reduced straight from the accumulators (two block reductions), the normalised tile is written once, 16 bytes per lane.
// LDS per workgroup: 77.6 KB (every buffer time-major, rows padded by 4 floats), see AM_* below.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AM_T 512
#define AM_WAVES 8
#define AM_BUF 1728                         // one [T][C+4] activation buffer of the Res1d groups (48 x 36 is the largest)
#define AM_XIN 0                            // [48][20] (14 input channels padded to 16)
#define AM_O0 960                           // o0 o1 o2 o3 ta tb tc
#define AM_FA (AM_O0 + 7 * AM_BUF)          // [48][132] FPN level 0
#define AM_RED (AM_FA + 6336)
#define AM_LDS_FLOATS (AM_RED + 32)
#define AM_MAXT 3                           // output tiles per wave (24 tiles of the 128 x 48 layers / 8 waves)

struct AmRes { const u32 *c1, *c2, *ds; const float *g1, *b1, *g2, *b2, *gd, *bd; };
struct AmLat { const u32 *w; const float *g, *b; };
struct AmW {                                // packed conv fragments + GroupNorm affine, same order as ActorW
  AmRes res[9];
  AmLat lat[4];
};

__device__ __forceinline__ float am_block_sum(float v, float *red, int flip) {
  v = wave_sum(v);
  float *r = red + (flip & 1) * 16;
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < AM_WAVES; ++w) s += r[w];
  return s;
}

// x = hi + mid + lo exactly (3 x 8 significand bits): the bf16 parts of two values packed per dword
__device__ __forceinline__ void am_split(const f32x4 &x0, const f32x4 &x1, u32x4 &h, u32x4 &m, u32x4 &l, int np) {
  h[0] = pk_bf16(x0[0], x0[1]); h[1] = pk_bf16(x0[2], x0[3]);
  h[2] = pk_bf16(x1[0], x1[1]); h[3] = pk_bf16(x1[2], x1[3]);
  if (np == 1) return;
  f32x4 r0, r1;
  r0[0] = x0[0] - bf_lo_f32(h[0]); r0[1] = x0[1] - bf_hi_f32(h[0]); r0[2] = x0[2] - bf_lo_f32(h[1]); r0[3] = x0[3] - bf_hi_f32(h[1]);
  r1[0] = x1[0] - bf_lo_f32(h[2]); r1[1] = x1[1] - bf_hi_f32(h[2]); r1[2] = x1[2] - bf_lo_f32(h[3]); r1[3] = x1[3] - bf_hi_f32(h[3]);
  m[0] = pk_bf16(r0[0], r0[1]); m[1] = pk_bf16(r0[2], r0[3]);
  m[2] = pk_bf16(r1[0], r1[1]); m[3] = pk_bf16(r1[2], r1[3]);
  if (np == 3) return;
  l[0] = pk_bf16(r0[0] - bf_lo_f32(m[0]), r0[1] - bf_hi_f32(m[0])); l[1] = pk_bf16(r0[2] - bf_lo_f32(m[1]), r0[3] - bf_hi_f32(m[1]));
  l[2] = pk_bf16(r1[0] - bf_lo_f32(m[2]), r1[1] - bf_hi_f32(m[2])); l[3] = pk_bf16(r1[2] - bf_lo_f32(m[3]), r1[3] - bf_hi_f32(m[3]));
}

// raw conv tiles of this wave: acc[i] = tile (wave + 8 i) of the (Cout/16) x ceil(Tout/16) grid, time tile fastest.
// NP = number of partial products per term: 6 = both operands split three ways (hi.hi + hi.mid + mid.hi + mid.mid + hi.lo +
// lo.hi: ~2^-24 relative, fp32-class), 3 = two-way split (hi.hi + hi.mid + mid.hi: ~2^-16), 1 = plain bf16 operands.  The
// leading products and the corrections run in two accumulators (two independent MFMA chains), summed at the end.
template <int NP, int LGC /*log2 Cin_pad*/, int KSZ, int STRIDE>
__device__ __forceinline__ void am_conv(const float *in, int Tin, const u32 *__restrict__ Wf, int Cout, int Tout,
                                        f32x4 (&acc)[AM_MAXT]) {
  constexpr int CP = 1 << LGC, LD = CP + 4, PAD = (KSZ - 1) / 2;
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
      f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
      if (dk < KSZ && t < Tout && row >= 0 && row < Tin) {
        x0 = *(const f32x4 *)(in + row * LD + ci);
        x1 = *(const f32x4 *)(in + row * LD + ci + 4);
      }
      u32x4 bh, bm, bl;
      am_split(x0, x1, bh, bm, bl, NP);
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

// GroupNorm(1 group) over the Cout x Tout outputs held in the accumulators, per-channel affine, optional residual (time-major
// LDS image with the same row length), optional ReLU; written to `out` ([Tout][Cout + 4]).  With `gout` (final layer) only
// the last time column is kept and goes to global memory.  Ends with a barrier.
__device__ __forceinline__ void am_gn(f32x4 (&acc)[AM_MAXT], int Cout, int Tout, const float *__restrict__ g,
                                      const float *__restrict__ b, const float *resid, bool relu, float *out,
                                      float *red, float *__restrict__ gout = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int ntt = (Tout + 15) >> 4;
  const int tiles = (Cout >> 4) * ntt;
  const int LD = Cout + 4;
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
    if (resid) {
      const f32x4 rr = *(const f32x4 *)(resid + t * LD + co);
      y += rr;
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
    }
    if (gout) {
      if (t == Tout - 1) *(f32x4 *)(gout + co) = y;
    } else {
      *(f32x4 *)(out + t * LD + co) = y;
    }
  }
  __syncthreads();
}

// Res1d (layers.py:175-188): in [Tin][Cin_pad+4] -> out [Tout][Cout+4]; scratch t1, t2
template <int NP, int LGCI, int LGCO, int STRIDE, bool DS>
__device__ __forceinline__ void am_res(const float *in, int Tin, const AmRes &W, float *out, float *t1, float *t2,
                                       float *red, float *__restrict__ gout = nullptr) {
  constexpr int Cout = 1 << LGCO;
  const int Tout = Tin / STRIDE;
  f32x4 acc[AM_MAXT];
  am_conv<NP, LGCI, 3, STRIDE>(in, Tin, W.c1, Cout, Tout, acc);
  am_gn(acc, Cout, Tout, W.g1, W.b1, nullptr, true, t1, red);
  const float *resid = in;
  if (DS) {
    am_conv<NP, LGCI, 1, STRIDE>(in, Tin, W.ds, Cout, Tout, acc);
    am_gn(acc, Cout, Tout, W.gd, W.bd, nullptr, false, t2, red);
    resid = t2;
  }
  am_conv<NP, LGCO, 3, 1>(t1, Tout, W.c2, Cout, Tout, acc);
  am_gn(acc, Cout, Tout, W.g2, W.b2, resid, true, out, red, gout);
}

// lateral conv + GroupNorm of FPN level g (src [T][C+4]) plus the x2 linear upsample (align_corners = False) of the level above
// (`up`, [T/2][132]; null at the top level) -> dst [T][132]   (network.py:55-58)
template <int NP, int LGC>
__device__ __forceinline__ void am_lateral(const float *src, int T, const AmLat &W, const float *up, float *dst,
                                           float *red) {
  f32x4 acc[AM_MAXT];
  am_conv<NP, LGC, 3, 1>(src, T, W.w, 128, T, acc);
  am_gn(acc, 128, T, W.g, W.b, nullptr, false, dst, red);
  if (up) {
    const int Th = T >> 1;
    for (int i = threadIdx.x; i < T * 32; i += AM_T) {
      const int t = i >> 5, c = (i & 31) * 4;
      float sp = (t + 0.5f) * 0.5f - 0.5f;
      sp = sp < 0.f ? 0.f : sp;
      const int i0 = (int)sp;
      const int i1 = i0 + 1 < Th ? i0 + 1 : Th - 1;
      const float l1 = sp - (float)i0;
      const f32x4 a0 = *(const f32x4 *)(up + i0 * 132 + c), a1 = *(const f32x4 *)(up + i1 * 132 + c);
      f32x4 d = *(const f32x4 *)(dst + t * 132 + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] += (1.0f - l1) * a0[e] + l1 * a1[e];
      *(f32x4 *)(dst + t * 132 + c) = d;
    }
    __syncthreads();
  }
}

template <int NP>
__global__ __launch_bounds__(AM_T, 2) void k_actor_mfma(const float *__restrict__ actors /*[A,14,48]*/, int n_actors,
                                                        float *__restrict__ out /*[A,128]*/, AmW W) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *xin = sm + AM_XIN;
  float *o0 = sm + AM_O0, *o1 = o0 + AM_BUF, *o2 = o1 + AM_BUF, *o3 = o2 + AM_BUF;
  float *ta = o3 + AM_BUF, *tb = ta + AM_BUF, *tc = tb + AM_BUF;
  float *fa = sm + AM_FA, *red = sm + AM_RED;
  const int tid = threadIdx.x;
  const int a = blockIdx.x;
  if (a >= n_actors) return;
  for (int i = tid; i < 48 * 20; i += AM_T) {
    const int t = i / 20, c = i - t * 20;
    xin[i] = c < 14 ? actors[((size_t)a * 14 + c) * 48 + t] : 0.f;
  }
  __syncthreads();
  am_res<NP, 4, 5, 1, true>(xin, 48, W.res[0], ta, tb, tc, red);
  am_res<NP, 5, 5, 1, false>(ta, 48, W.res[1], o0, tb, tc, red);
  am_res<NP, 5, 6, 2, true>(o0, 48, W.res[2], ta, tb, tc, red);
  am_res<NP, 6, 6, 1, false>(ta, 24, W.res[3], o1, tb, tc, red);
  am_res<NP, 6, 7, 2, true>(o1, 24, W.res[4], ta, tb, tc, red);
  am_res<NP, 7, 7, 1, false>(ta, 12, W.res[5], o2, tb, tc, red);
  am_res<NP, 7, 8, 2, true>(o2, 12, W.res[6], ta, tb, tc, red);
  am_res<NP, 8, 8, 1, false>(ta, 6, W.res[7], o3, tb, tc, red);
  // FPN top-down: level 3 -> ta, level 2 -> tb, level 1 -> o2..o3 (dead by then), level 0 -> fa
  am_lateral<NP, 8>(o3, 6, W.lat[3], nullptr, ta, red);
  am_lateral<NP, 7>(o2, 12, W.lat[2], ta, tb, red);
  am_lateral<NP, 6>(o1, 24, W.lat[1], tb, o2, red);
  am_lateral<NP, 5>(o0, 48, W.lat[0], o2, fa, red);
  // output Res1d(128, 128) at T = 48 (network.py:60); only the last time column is kept.  conv1's output goes to o0..o3 (dead)
  {
    const AmRes &R = W.res[8];
    f32x4 acc[AM_MAXT];
    am_conv<NP, 7, 3, 1>(fa, 48, R.c1, 128, 48, acc);
    am_gn(acc, 128, 48, R.g1, R.b1, nullptr, true, o0, red);
    am_conv<NP, 7, 3, 1>(o0, 48, R.c2, 128, 48, acc);
    am_gn(acc, 128, 48, R.g2, R.b2, fa, true, o0, red, out + (size_t)a * 128);
  }
}
extern "C" size_t mind_actor_mfma_lds_bytes() { return (size_t)AM_LDS_FLOATS * sizeof(float); }
