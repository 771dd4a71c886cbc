// SceneDecoder, actor part (actor_proj 128 -> 384 -> 768, mode embedding, reg head 128 -> 128 -> 128 -> 40, Bezier evaluation) with
// its GEMMs on the bf16 MFMA, fp32-accurate by the three-way operand split of actor_mfma_kernels.hip (same fragment order, same
// LDS image of three bf16 planes per activation row).
//
// Reference semantics: planners/mind/networks/network.py:483-556 (SceneDecoder.forward, actor branch); same math as k_dec_actor
// (encdec_kernels.hip), which stays the fp32 (VALU) arithmetic.
//
// DM_RA agents per workgroup: a Linear is out[f][row] = sum_k W[f][k] X[row][k]: weights = A operand (packed fragments
// [m-tile][k-step][part][lane][4 dwords] from the host, pack_conv_frag with one tap), activation rows = B operand (one ds_read_b128
// per plane), C/D = 4 consecutive output features of one row.  LayerNorm is per row, i.e. across m-tiles: the GEMM writes
// bias-added fp32 rows, one wave per row normalises and writes the next layer's split image.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DM_T 512
#define DM_WAVES 8
#define DM_RA 16
#define DM_ROWS (DM_RA * 6)
#define DM_R1 0                                   // split images: xin [16] x RSD(128), h1 [16] x RSD(384), E / t1 / E2 [96] x RSD(128)
#define DM_R2 (DM_ROWS * AM_RSD(128))             // fp32 rows: h1 raw [16][388], h2 [16][772], t1 raw [96][132], prm [96][40]
#define DM_LDS_DWORDS (DM_R2 + DM_ROWS * 132)

struct DmW {
  const u32 *a0, *a3, *r0, *r3, *r6;              // packed fragments
  const float *a0b, *a0g, *a0be, *a3b, *a3g, *a3be, *r0b, *r0g, *r0be, *r3b, *r3g, *r3be, *r6b;
  const float *T, *Tp;                            // Bezier bases [60][8], [60][7]
};

// out[row][f] = bias[f] + sum_k W[f][k] in[row][k] for `rows` rows, f < n_out (m-tiles past n_out are zero rows of the packing)
template <int NP, int CP>
__device__ __forceinline__ void dm_gemm(const u32 *in, int rows, const u32 *__restrict__ Wf, int mts, const float *__restrict__ bias,
                                        int n_out, float *out, int ldo) {
  constexpr int KS = CP / 32, RSD = AM_RSD(CP), PLANE = CP / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int ntt = (rows + 15) >> 4;
  const int tiles = mts * ntt;
  for (int ti = wave; ti < tiles; ti += DM_WAVES) {
    const int mt = ti / ntt, nt = ti - mt * ntt;
    const int t = nt * 16 + r;
    const u32 *wp = Wf + (size_t)mt * KS * 768 + lane * 4;
    const u32 *bp = in + t * RSD + q * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, corr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 ah = *(const u32x4 *)(wp + (size_t)ks * 768);
      u32x4 am, al;
      if (NP >= 3) am = *(const u32x4 *)(wp + (size_t)ks * 768 + 256);
      if (NP == 6) al = *(const u32x4 *)(wp + (size_t)ks * 768 + 512);
      u32x4 bh = {0u, 0u, 0u, 0u}, bm = {0u, 0u, 0u, 0u}, bl = {0u, 0u, 0u, 0u};
      if (t < rows) {
        bh = *(const u32x4 *)(bp + ks * 16);
        if (NP >= 3) bm = *(const u32x4 *)(bp + ks * 16 + PLANE);
        if (NP == 6) bl = *(const u32x4 *)(bp + ks * 16 + 2 * PLANE);
      }
      acc = MFMA_BF(ah, bh, acc);
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
    if (NP >= 3) acc += corr;
    const int co = mt * 16 + q * 4;
    if (t < rows && co < n_out) {
      const f32x4 bb = *(const f32x4 *)(bias + co);
      *(f32x4 *)(out + t * ldo + co) = acc + bb;
    }
  }
}

__device__ __forceinline__ void dm_store_pair(u32 *p, int plane, float y0, float y1) {
  const u32 h = pk_bf16(y0, y1);
  const float r0 = y0 - bf_lo_f32(h), r1 = y1 - bf_hi_f32(h);
  const u32 m = pk_bf16(r0, r1);
  p[0] = h; p[plane] = m; p[2 * plane] = pk_bf16(r0 - bf_lo_f32(m), r1 - bf_hi_f32(m));
}

// LayerNorm (+ReLU) over the n features of each row of `raw` (n a multiple of 128), one wave per row: -> split image `outs`
// ([rows] x AM_RSD(n)), or in place (fp32) when outs is null
__device__ __forceinline__ void dm_ln(float *raw, int ld, int rows, int n, const float *__restrict__ g, const float *__restrict__ be,
                                      u32 *outs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int RSD = 3 * n / 2 + 4, PLANE = n >> 1;
  for (int r = wave; r < rows; r += DM_WAVES) {
    float *row = raw + r * ld;
    float s = 0.f;
    for (int c = 2 * lane; c < n; c += 128) s += row[c] + row[c + 1];
    const float mean = wave_sum(s) / (float)n;
    float v = 0.f;
    for (int c = 2 * lane; c < n; c += 128) {
      const float d0 = row[c] - mean, d1 = row[c + 1] - mean;
      v = fmaf(d0, d0, v); v = fmaf(d1, d1, v);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)n + 1e-5f);
    for (int c = 2 * lane; c < n; c += 128) {
      const float y0 = fmaxf((row[c] - mean) * rstd * g[c] + be[c], 0.f);
      const float y1 = fmaxf((row[c + 1] - mean) * rstd * g[c + 1] + be[c + 1], 0.f);
      if (outs) dm_store_pair(outs + r * RSD + (c >> 1), PLANE, y0, y1);
      else { row[c] = y0; row[c + 1] = y1; }
    }
  }
}

template <int NP>
__global__ __launch_bounds__(DM_T) void k_dec_actor_mfma(const float *__restrict__ x /*[tokens,128]*/, const int *__restrict__ actor_row,
                                                         const int *__restrict__ actor_scene, int n_actors,
                                                         const float *__restrict__ Cmode /*[B,6,128]*/,
                                                         const float *__restrict__ tgt /*[B,128]*/, float *__restrict__ reg,
                                                         float *__restrict__ vel, DmW W) {
  extern __shared__ __attribute__((aligned(16))) u32 dmu[];
  u32 *R1 = dmu + DM_R1;
  float *R2 = (float *)(dmu + DM_R2);
  const int tid = threadIdx.x;
  const int a0 = blockIdx.x * DM_RA;
  const int na = min(DM_RA, n_actors - a0);
  // fused actor tokens -> split image [16] x RSD(128)
  for (int i = tid; i < DM_RA * 64; i += DM_T) {
    const int r = i >> 6, c = (i & 63) * 2;
    float v0 = 0.f, v1 = 0.f;
    if (r < na) {
      const float *xr = x + (size_t)actor_row[a0 + r] * 128 + c;
      v0 = xr[0]; v1 = xr[1];
    }
    dm_store_pair(R1 + r * AM_RSD(128) + (c >> 1), 64, v0, v1);
  }
  __syncthreads();
  dm_gemm<NP, 128>(R1, DM_RA, W.a0, 24, W.a0b, 384, R2, 388);
  __syncthreads();
  dm_ln(R2, 388, DM_RA, 384, W.a0g, W.a0be, R1);
  __syncthreads();
  dm_gemm<NP, 384>(R1, DM_RA, W.a3, 48, W.a3b, 768, R2, 772);
  __syncthreads();
  dm_ln(R2, 772, DM_RA, 768, W.a3g, W.a3be, nullptr);
  __syncthreads();
  // embed = cls_embed + actor_embed (+ tgt on mode 0 only)  (network.py:506-510) -> split image [96] x RSD(128)
  for (int i = tid; i < DM_ROWS * 64; i += DM_T) {
    const int row = i >> 6, c = (i & 63) * 2;
    const int r = row / 6, k = row - r * 6;
    float v0 = 0.f, v1 = 0.f;
    if (r < na) {
      const int sc = actor_scene[a0 + r];
      const float *cm = Cmode + ((size_t)sc * 6 + k) * 128 + c;
      v0 = R2[r * 772 + k * 128 + c] + cm[0];
      v1 = R2[r * 772 + k * 128 + c + 1] + cm[1];
      if (k == 0) { v0 += tgt[(size_t)sc * 128 + c]; v1 += tgt[(size_t)sc * 128 + c + 1]; }
    }
    dm_store_pair(R1 + row * AM_RSD(128) + (c >> 1), 64, v0, v1);
  }
  __syncthreads();
  const int rows = na * 6;
  dm_gemm<NP, 128>(R1, rows, W.r0, 8, W.r0b, 128, R2, 132);
  __syncthreads();
  dm_ln(R2, 132, rows, 128, W.r0g, W.r0be, R1);
  __syncthreads();
  dm_gemm<NP, 128>(R1, rows, W.r3, 8, W.r3b, 128, R2, 132);
  __syncthreads();
  dm_ln(R2, 132, rows, 128, W.r3g, W.r3be, R1);
  __syncthreads();
  // 128 -> 40 = 8 control points x (x, y, sx, sy, rho)
  float (*prm)[40] = (float (*)[40])R2;
  dm_gemm<NP, 128>(R1, rows, W.r6, 3, W.r6b, 40, R2, 40);
  __syncthreads();
  // Bezier evaluation (network.py:515-523,545): pos = T P, cov = exp(T S), vel = Tp dP / 6
  for (int i = tid; i < rows * 60; i += DM_T) {
    const int r = i / 60, t = i - r * 60;
    const int a = a0 + r / 6, k = r % 6;
    float o5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < 8; ++c) {
      const float tv = W.T[t * 8 + c];
#pragma unroll
      for (int d = 0; d < 5; ++d) o5[d] = fmaf(tv, prm[r][c * 5 + d], o5[d]);
    }
    float v2[2] = {0.f, 0.f};
    for (int c = 0; c < 7; ++c) {
      const float tv = W.Tp[t * 7 + c];
      v2[0] = fmaf(tv, prm[r][(c + 1) * 5 + 0] - prm[r][c * 5 + 0], v2[0]);
      v2[1] = fmaf(tv, prm[r][(c + 1) * 5 + 1] - prm[r][c * 5 + 1], v2[1]);
    }
    float *ro = reg + (((size_t)a * 6 + k) * 60 + t) * 5;
    ro[0] = o5[0]; ro[1] = o5[1]; ro[2] = expf(o5[2]); ro[3] = expf(o5[3]); ro[4] = expf(o5[4]);
    float *vo = vel + (((size_t)a * 6 + k) * 60 + t) * 2;
    vo[0] = v2[0] / 6.0f; vo[1] = v2[1] / 6.0f;
  }
}
extern "C" size_t mind_dec_actor_mfma_lds_bytes() { return (size_t)DM_LDS_DWORDS * sizeof(u32); }
