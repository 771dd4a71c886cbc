// ActorNet (Res1d groups + FPN + output Res1d) on the fp32 MFMA (v_mfma_f32_16x16x4_f32): plain fp32 operands, fp32 accumulate -- the
// reference's own arithmetic class -- with NACT actors per workgroup.
//
// Reference semantics: planners/mind/networks/network.py:20-61 (ActorNet), layers.py:127-188 (Conv1d / Res1d with GroupNorm(1 group));
// same math as k_actor_net (encdec_kernels.hip: fp32 VALU, one actor per workgroup, bound by its LDS operand traffic: 269 us per launch at
// demo size) and k_actor_mfma (actor_mfma_kernels.hip: operands split into three bf16 planes, six products per term).
//
// Why a third kernel.  (1) "exact fp32" (MIND_PAIR_PREC=f32) ran the VALU kernel: twice the time of the split-bf16 one.  The fp32 MFMA has the
// VALU's FLOP rate but takes its 16 x 16 operand tile from registers once instead of one LDS operand per FMA.  (2) The split-bf16 kernel holds
// ONE actor per workgroup (its three bf16 planes of every activation fill 110 KB of LDS) and streams 9.1 MB of weight fragments per actor through
// one CU's 64 B/clk L2 port; more than half of those bytes feed the 256-channel layers, which have SIX time steps -- 6 of the 16 columns of an
// MFMA tile.  fp32 activations are 4 bytes instead of 6: two actors fit a workgroup (155 KB), the 6-step layers then fill 12 of 16 columns with the
// same weight fragments, and a weight is 4 bytes instead of 6: 4.4 MB per actor instead of 9.1 at full-tree scale.
//
// A Conv1d is a GEMM  out[co][n] = sum_k Wm[co][k] X[k][n],  k = dk * Cin_pad + ci,  n = actor * Tout + t,  X[k][n] = in_actor[ci][t * stride + dk - pad]:
//   * M = output channels: the weights are the A operand, packed on the host (pack_conv_f32, mind_hip.hip) as [m-tile][k-group of 16][lane 64][4 floats]:
//     lane (r = lane & 15, q = lane >> 4) holds row co = 16 mt + r, k = 16 g + 4 q + (0..3): one 16-byte load per lane and FOUR MFMAs
//     (MFMA m of the group contracts the k-slots 4 q + m of its four lane quarters: any assignment of the 16 k-values to (MFMA, quarter) is
//     the same sum);
//   * N = (actor, time): the activations are the B operand: every tensor lives in LDS as a time-major fp32 image per actor, row t = [C floats][4 pad];
//     the lane's four k-slots -- four consecutive input channels at one tap -- are one ds_read_b128; rows outside [0, Tin) are zeros;
//   * C/D: lane holds column n = 16 nt + (lane & 15), channels 16 mt + 4 (lane >> 4) + (0..3): GroupNorm statistics per actor straight from
//     the accumulators, the normalised tile (+ residual, + the FPN's upsampled upper level, ReLU) written as one 16-byte store per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AF_T 1024
#define AF_WAVES 16
#define AF_MAXT 3                           // output tiles per wave: 48 tiles of the 128-channel T = 48 layers with two actors / 16 waves
#define AF_RS(C) ((C) + 4)                  // image row length in floats
#define AF_RING(NACT) ((NACT) == 1 ? 8 : 4) // weight-fragment groups in flight per wave (two actors per workgroup leave fewer registers)
#define AF_BUF 1728                         // one image of the Res1d groups: 48 x AF_RS(32) is the largest
#define AF_XIN 0                            // [48] x AF_RS(16)
#define AF_O0 960                           // o0 o1 o2 o3 ta tb tc
#define AF_FA (AF_O0 + 7 * AF_BUF)          // [48] x AF_RS(128): FPN level 0
#define AF_ACT (AF_FA + 48 * AF_RS(128))    // floats per actor: 19 392 (77.6 KB)
#define AF_LDS_FLOATS(NACT) ((NACT) * AF_ACT + 128)

struct AfRes { const float *c1, *c2, *ds; const float *g1, *b1, *g2, *b2, *gd, *bd; };
struct AfLat { const float *w; const float *g, *b; };
struct AfW {                                // packed conv fragments + GroupNorm affine, same order as ActorW / AmW
  AfRes res[9];
  AfLat lat[4];
};

// sums of NACT values over the workgroup (one per actor); `flip` alternates the scratch so that two reductions need no barrier in between
template <int NACT>
__device__ __forceinline__ void af_block_sum(float (&v)[NACT], float *red, int flip) {
  float *r = red + (flip & 1) * (AF_WAVES * 2);
#pragma unroll
  for (int e = 0; e < NACT; ++e) {
    const float s = wave_sum(v[e]);
    if ((threadIdx.x & 63) == 0) r[(threadIdx.x >> 6) * 2 + e] = s;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < NACT; ++e) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < AF_WAVES; ++w) s += r[w * 2 + e];
    v[e] = s;
  }
}

// raw conv tiles of this wave: acc[i] = tile (wave + 16 i) of the (Cout / 16) x ceil(NACT Tout / 16) grid, column tile fastest (the waves
// that share an m-tile run side by side: their weight-fragment loads meet in the CU's L1)
template <int NACT, int LGC /*log2 Cin_pad*/, int KSZ, int STRIDE>
__device__ __forceinline__ void af_conv(const float *in /*actor 0's image; actor e at + e AF_ACT*/, int Tin, const float *__restrict__ Wf, int Cout, int Tout,
                                        f32x4 (&acc)[AF_MAXT]) {
  constexpr int CP = 1 << LGC, RS = AF_RS(CP), PAD = (KSZ - 1) / 2;
  constexpr int G = KSZ * CP / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int ntt = (NACT * Tout + 15) >> 4;
  const int tiles = (Cout >> 4) * ntt;
#pragma unroll
  for (int i = 0; i < AF_MAXT; ++i) {
    acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ti = wave + AF_WAVES * i;
    if (ti >= tiles) continue;
    const int mt = ti / ntt, nt = ti - mt * ntt;
    const int n = nt * 16 + r;
    const int e = NACT == 1 ? 0 : (n >= Tout ? 1 : 0), t = n - e * Tout;
    const bool col = n < NACT * Tout;
    const float *img = in + e * AF_ACT;
    const f32x4 *wp = (const f32x4 *)(Wf + (size_t)mt * G * 256) + lane;
    // the weight fragments travel in a register ring, D groups ahead of their MFMAs (left to itself the compiler requests ONE group ahead and
    // waits for it: a full L2 round trip per 128 cycles of matrix pipe, 226 us per launch at demo size instead of 150)
    constexpr int D = G < AF_RING(NACT) ? G : AF_RING(NACT);
    f32x4 ar[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ar[d] = wp[(size_t)d * 64];
    for (int g0 = 0; g0 < G; g0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int g = g0 + d;
        if (g >= G) break;
        const f32x4 a = ar[d];
        if (g + D < G) ar[d] = wp[(size_t)(g + D) * 64];
        const int k0 = g * 16;
        const int dk = k0 >> LGC, ci = (k0 & (CP - 1)) + 4 * q;
        const int row = t * STRIDE + dk - PAD;
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (col && row >= 0 && row < Tin) b = *(const f32x4 *)(img + row * RS + ci);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc[i], 0, 0, 0);
      }
    }
  }
}

// GroupNorm(1 group) per actor over the Cout x Tout outputs held in the accumulators, per-channel affine, optional residual (an image with
// the same row length), optional x2 linear upsample (align_corners = False) of `up` (the level above, [Tout / 2] x AF_RS(128)) added, optional
// ReLU.  Output: the image `outs` ([Tout] x AF_RS(Cout); actor e at + e AF_ACT like every image pointer), or `gout` (final layer: only the last
// time column, to global memory).  Ends with a barrier.
template <int NACT, bool FINAL = false>
__device__ __forceinline__ void af_gn(f32x4 (&acc)[AF_MAXT], int Cout, int Tout, const float *__restrict__ g, const float *__restrict__ b,
                                      const float *resid, const float *up, bool relu, float *outs, float *gout, int a0, int n_actors,
                                      float *red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int ntt = (NACT * Tout + 15) >> 4;
  const int tiles = (Cout >> 4) * ntt;
  const int RS = AF_RS(Cout);
  f32x4 gg[AF_MAXT], bb[AF_MAXT];
  float s[NACT];
#pragma unroll
  for (int e = 0; e < NACT; ++e) s[e] = 0.f;
#pragma unroll
  for (int i = 0; i < AF_MAXT; ++i) {
    const int ti = wave + AF_WAVES * i;
    if (ti >= tiles) continue;
    const int mt = ti / ntt, nt = ti - mt * ntt;
    gg[i] = *(const f32x4 *)(g + mt * 16 + q * 4);
    bb[i] = *(const f32x4 *)(b + mt * 16 + q * 4);
    const int n = nt * 16 + r;
    if (n < NACT * Tout) {
      const float v = (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
      if (NACT == 1 || n < Tout) s[0] += v; else s[NACT - 1] += v;
    }
  }
  const float cnt = (float)(Cout * Tout);
  af_block_sum<NACT>(s, red, 0);
  float mean[NACT], v2[NACT];
#pragma unroll
  for (int e = 0; e < NACT; ++e) { mean[e] = s[e] / cnt; v2[e] = 0.f; }
#pragma unroll
  for (int i = 0; i < AF_MAXT; ++i) {
    const int ti = wave + AF_WAVES * i;
    if (ti >= tiles) continue;
    const int nt = ti % ntt;
    const int n = nt * 16 + r;
    if (n < NACT * Tout) {
      const int e = (NACT == 1 || n < Tout) ? 0 : NACT - 1;
      float v = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) { const float d = acc[i][c] - mean[e]; v = fmaf(d, d, v); }
      v2[e] += v;
    }
  }
  af_block_sum<NACT>(v2, red, 1);
  float rstd[NACT];
#pragma unroll
  for (int e = 0; e < NACT; ++e) rstd[e] = 1.0f / sqrtf(v2[e] / cnt + 1e-5f);
#pragma unroll
  for (int i = 0; i < AF_MAXT; ++i) {
    const int ti = wave + AF_WAVES * i;
    if (ti >= tiles) continue;
    const int mt = ti / ntt, nt = ti - mt * ntt;
    const int n = nt * 16 + r, co = mt * 16 + q * 4;
    if (n >= NACT * Tout) continue;
    const int e = (NACT == 1 || n < Tout) ? 0 : NACT - 1, t = n - e * Tout;
    f32x4 y;
#pragma unroll
    for (int c = 0; c < 4; ++c) y[c] = (acc[i][c] - mean[e]) * rstd[e] * gg[i][c] + bb[i][c];
    if (resid) y += *(const f32x4 *)(resid + e * AF_ACT + t * RS + co);
    if (up) {
      const int Th = Tout >> 1;
      float sp = (t + 0.5f) * 0.5f - 0.5f;
      sp = sp < 0.f ? 0.f : sp;
      const int i0 = (int)sp;
      const int i1 = i0 + 1 < Th ? i0 + 1 : Th - 1;
      const float l1 = sp - (float)i0;
      const float *ue = up + e * AF_ACT;
      const f32x4 u0 = *(const f32x4 *)(ue + i0 * AF_RS(128) + co), u1 = *(const f32x4 *)(ue + i1 * AF_RS(128) + co);
#pragma unroll
      for (int c = 0; c < 4; ++c) y[c] += (1.0f - l1) * u0[c] + l1 * u1[c];
    }
    if (relu) {
#pragma unroll
      for (int c = 0; c < 4; ++c) y[c] = fmaxf(y[c], 0.f);
    }
    if (FINAL) {
      if (t == Tout - 1 && a0 + e < n_actors) *(f32x4 *)(gout + (size_t)(a0 + e) * 128 + co) = y;
    } else {
      *(f32x4 *)(outs + e * AF_ACT + t * RS + co) = y;
    }
  }
  __syncthreads();
}

// Res1d (layers.py:175-188): in [Tin] x AF_RS(Cin_pad) -> out [Tout] x AF_RS(Cout); scratch t1, t2
template <int NACT, int LGCI, int LGCO, int STRIDE, bool DS>
__device__ __forceinline__ void af_res(const float *in, int Tin, const AfRes &W, float *out, float *t1, float *t2, float *red) {
  constexpr int Cout = 1 << LGCO;
  const int Tout = Tin / STRIDE;
  f32x4 acc[AF_MAXT];
  af_conv<NACT, LGCI, 3, STRIDE>(in, Tin, W.c1, Cout, Tout, acc);
  af_gn<NACT>(acc, Cout, Tout, W.g1, W.b1, nullptr, nullptr, true, t1, nullptr, 0, 0, red);
  const float *resid = in;
  if (DS) {
    af_conv<NACT, LGCI, 1, STRIDE>(in, Tin, W.ds, Cout, Tout, acc);
    af_gn<NACT>(acc, Cout, Tout, W.gd, W.bd, nullptr, nullptr, false, t2, nullptr, 0, 0, red);
    resid = t2;
  }
  af_conv<NACT, LGCO, 3, 1>(t1, Tout, W.c2, Cout, Tout, acc);
  af_gn<NACT>(acc, Cout, Tout, W.g2, W.b2, resid, nullptr, true, out, nullptr, 0, 0, red);
}

// FPN level (network.py:55-58): lateral conv + GroupNorm of `src` plus the upsampled level above (`up`; null at the top) -> `dst` ([T] x AF_RS(128))
template <int NACT, int LGC>
__device__ __forceinline__ void af_lateral(const float *src, int T, const AfLat &W, const float *up, float *dst, float *red) {
  f32x4 acc[AF_MAXT];
  af_conv<NACT, LGC, 3, 1>(src, T, W.w, 128, T, acc);
  af_gn<NACT>(acc, 128, T, W.g, W.b, nullptr, up, false, dst, nullptr, 0, 0, red);
}

template <int NACT>
__global__ __launch_bounds__(AF_T) void k_actor_f32(const float *__restrict__ actors /*[A,14,48]*/, int n_actors, float *__restrict__ out /*[A,128]*/, AfW W) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  float *xin = smf + AF_XIN;
  float *o0 = smf + AF_O0, *o1 = o0 + AF_BUF, *o2 = o1 + AF_BUF, *o3 = o2 + AF_BUF;
  float *ta = o3 + AF_BUF, *tb = ta + AF_BUF, *tc = tb + AF_BUF;
  float *fa = smf + AF_FA;
  float *red = smf + NACT * AF_ACT;
  const int tid = threadIdx.x;
  const int a0 = blockIdx.x * NACT;
  if (a0 >= n_actors) return;
  // input [14][48] -> image [48] x AF_RS(16), channels 14, 15 zero; an actor past the end (odd count, two per workgroup) is all zeros
  for (int i = tid; i < NACT * 48 * 16; i += AF_T) {
    const int e = i / (48 * 16), t = (i / 16) % 48, c = i & 15;
    const bool have = c < 14 && a0 + e < n_actors;
    xin[e * AF_ACT + t * AF_RS(16) + c] = have ? actors[((size_t)(a0 + e) * 14 + c) * 48 + t] : 0.f;
  }
  __syncthreads();
  af_res<NACT, 4, 5, 1, true>(xin, 48, W.res[0], ta, tb, tc, red);
  af_res<NACT, 5, 5, 1, false>(ta, 48, W.res[1], o0, tb, tc, red);
  af_res<NACT, 5, 6, 2, true>(o0, 48, W.res[2], ta, tb, tc, red);
  af_res<NACT, 6, 6, 1, false>(ta, 24, W.res[3], o1, tb, tc, red);
  af_res<NACT, 6, 7, 2, true>(o1, 24, W.res[4], ta, tb, tc, red);
  af_res<NACT, 7, 7, 1, false>(ta, 12, W.res[5], o2, tb, tc, red);
  af_res<NACT, 7, 8, 2, true>(o2, 12, W.res[6], ta, tb, tc, red);
  af_res<NACT, 8, 8, 1, false>(ta, 6, W.res[7], o3, tb, tc, red);
  // FPN top-down: level 3 -> ta, 2 -> tb, 1 -> o2..o3 (dead by then: 24 x AF_RS(128) = 3 168 floats <= 2 AF_BUF), 0 -> fa
  af_lateral<NACT, 8>(o3, 6, W.lat[3], nullptr, ta, red);
  af_lateral<NACT, 7>(o2, 12, W.lat[2], ta, tb, red);
  af_lateral<NACT, 6>(o1, 24, W.lat[1], tb, o2, red);
  af_lateral<NACT, 5>(o0, 48, W.lat[0], o2, fa, red);
  // output Res1d(128, 128) at T = 48 (network.py:60); only the last time column is kept.  conv1's output goes to o0..o3 (dead: 48 x AF_RS(128)
  // = 6 336 floats <= 4 AF_BUF)
  {
    const AfRes &R = W.res[8];
    f32x4 acc[AF_MAXT];
    af_conv<NACT, 7, 3, 1>(fa, 48, R.c1, 128, 48, acc);
    af_gn<NACT>(acc, 128, 48, R.g1, R.b1, nullptr, nullptr, true, o0, nullptr, 0, 0, red);
    af_conv<NACT, 7, 3, 1>(o0, 48, R.c2, 128, 48, acc);
    af_gn<NACT, true>(acc, 128, 48, R.g2, R.b2, fa, nullptr, true, nullptr, out, a0, n_actors, red);
  }
}
extern "C" size_t mind_actor_f32_lds_bytes(int nact) { return (size_t)AF_LDS_FLOATS(nact) * sizeof(float); }
