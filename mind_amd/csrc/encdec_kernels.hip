// Actor / lane polyline encoders and the K-mode Bezier scene decoder for gfx950 (fp32).
//
// Reference semantics:
//   ActorNet      planners/mind/networks/network.py:12-61, layers.py:36-60 (Conv1d), :140-188 (Res1d)
//   LaneNet       network.py:64-121
//   SceneDecoder  network.py:343-556 (bezier branch :408-421,:514-523; basis :449-464)
//
// Each kernel keeps a whole instance (one agent's FPN pyramid / two lane polylines / one scene's
// mode tokens / four agents' mode embeddings) resident in LDS and streams the weights, stored
// transposed ([in][out]) so a wave reads consecutive floats, from L2.  These are <3% of the
// predictor FLOPs; the dominant kernel is fusion_kernels.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NT 256

__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 32, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}

__device__ __forceinline__ float block_sum(float v, float *red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// out[r][o] = b[o] + sum_k in[r][k] WT[k][o],  r < R (compile time), o < NOUT; K % 4 == 0.
// `in` rows must be 16-byte aligned (ldi % 4 == 0).  These layers are latency-bound on streaming the
// weights from L2, so when NOUT < blockDim the K range is split over KP thread groups (partials
// reduced through the LDS scratch `part`, >= part_floats floats) and every thread keeps 16 weight
// loads in flight.  Contains __syncthreads(): call from uniform control flow; caller syncs after.
#ifndef DB
#define DB 16   // weight loads in flight per thread (32 was measured slower: register pressure at 1024 threads)
#endif
// Keeps the LDS reads of one input row next to the FMAs that consume them: without it the scheduler hoists the reads
// of ALL R rows above the first FMA (R x DB live values) and spills hundreds of registers at 1024 threads per workgroup.
#define DENSE_ROW_FENCE() __builtin_amdgcn_sched_barrier(0)
// Few input rows (R <= 8): a thread owns FOUR consecutive outputs, so every weight load is 16 bytes wide, and the K
// range is split over as many thread groups as the `part` scratch holds -- four times fewer dependent load batches per
// thread than one output per thread.  out[r][o] = b[o] + sum over the K parts (ascending k inside a part).
#define DBQ 8
template <int R>
__device__ __forceinline__ void dense_quads(const float *in, int ldi, int K, const float *__restrict__ WT,
                                            const float *__restrict__ b, int NOUT, float *out, int ldo,
                                            float *part, int part_floats) {
  const int nthr = blockDim.x;
  const int NQ = NOUT >> 2;
  int KP = 1;
  while (KP * 2 * NQ <= nthr && KP * 2 * R * NOUT <= part_floats && KP * 2 <= (K >> 2)) KP *= 2;
  const int Kc = ((K / 4 + KP - 1) / KP) * 4;
  for (int item = threadIdx.x; item < KP * NQ; item += nthr) {      // KP == 1: more quads than threads is allowed
    const int kp = item / NQ, o = (item % NQ) * 4;
    float acc[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = (KP == 1 && b) ? b[o + c] : 0.f;
    const int k0 = kp * Kc, k1 = min(K, k0 + Kc);
    int k = k0;
    for (; k + DBQ <= k1; k += DBQ) {
      f32x4 w[DBQ];
#pragma unroll
      for (int q = 0; q < DBQ; ++q) w[q] = *(const f32x4 *)(WT + (size_t)(k + q) * NOUT + o);
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int q4 = 0; q4 < DBQ / 4; ++q4) {
          const f32x4 x = *(const f32x4 *)(in + r * ldi + k + 4 * q4);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(w[4 * q4 + j][c], x[j], acc[r][c]);
        }
        DENSE_ROW_FENCE();
      }
    }
    for (; k < k1; k += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 w = *(const f32x4 *)(WT + (size_t)(k + j) * NOUT + o);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float x = in[r * ldi + k + j];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(w[c], x, acc[r][c]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (KP == 1) out[r * ldo + o + c] = acc[r][c];
        else part[(kp * R + r) * NOUT + o + c] = acc[r][c];
      }
  }
  if (KP == 1) return;
  __syncthreads();
  for (int i = threadIdx.x; i < R * NOUT; i += nthr) {
    const int r = i / NOUT, oo = i % NOUT;
    float v = b ? b[oo] : 0.f;
    for (int q = 0; q < KP; ++q) v += part[(q * R + r) * NOUT + oo];
    out[r * ldo + oo] = v;
  }
}

#ifndef DT
#define DT 1024   // threads of the latency-bound dense kernels (K-split over thread groups)
#endif
// ---- a dense_quads stage spread over the G workgroups of one scene (k_dec_scene_mw): workgroup `wg` owns the output quads
// [NQ wg / G, NQ (wg + 1) / G) with EVERY K part of them -- the K split (KP, Kc) is the one dense_quads takes in a 1 024-thread workgroup that
// owns all quads, and an item (K part, quad) is computed by one thread exactly as there, so every output is the same bits.  What changes is
// the traffic through the CU's L2 port (a stage of 0.8-1.2 MB of weights runs at ~60 % of one port; an eighth of it per CU is latency only).
// On return the workgroup's range of out[r][o] is valid in LDS.
template <int R>
__device__ __forceinline__ void dense_quads_part(const float *in, int ldi, int K, const float *__restrict__ WT, const float *__restrict__ b, int NOUT,
                                                 float *out, int ldo, float *part, int part_floats, int wg, int G) {
  const int nthr = blockDim.x;
  const int NQ = NOUT >> 2;
  int KP = 1;
  while (KP * 2 * NQ <= DT && KP * 2 * R * NOUT <= part_floats && KP * 2 <= (K >> 2)) KP *= 2;        // (DT: the one-workgroup kernel's thread count)
  const int Kc = ((K / 4 + KP - 1) / KP) * 4;
  const int q_lo = NQ * wg / G, q_hi = NQ * (wg + 1) / G, NQg = q_hi - q_lo;
  for (int item = threadIdx.x; item < KP * NQg; item += nthr) {
    const int kp = item / NQg, o = (q_lo + item % NQg) * 4;
    float acc[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = (KP == 1 && b) ? b[o + c] : 0.f;
    const int k0 = kp * Kc, k1 = min(K, k0 + Kc);
    int k = k0;
    for (; k + DBQ <= k1; k += DBQ) {
      f32x4 w[DBQ];
#pragma unroll
      for (int q = 0; q < DBQ; ++q) w[q] = *(const f32x4 *)(WT + (size_t)(k + q) * NOUT + o);
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int q4 = 0; q4 < DBQ / 4; ++q4) {
          const f32x4 x = *(const f32x4 *)(in + r * ldi + k + 4 * q4);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(w[4 * q4 + j][c], x[j], acc[r][c]);
        }
        DENSE_ROW_FENCE();
      }
    }
    for (; k < k1; k += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 w = *(const f32x4 *)(WT + (size_t)(k + j) * NOUT + o);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float x = in[r * ldi + k + j];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(w[c], x, acc[r][c]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (KP == 1) out[r * ldo + o + c] = acc[r][c];
        else part[(kp * R + r) * NOUT + o + c] = acc[r][c];
      }
  }
  if (KP == 1) return;
  __syncthreads();
  const int o_lo = q_lo * 4, no = NQg * 4;
  for (int i = threadIdx.x; i < R * no; i += nthr) {
    const int r = i / no, oo = o_lo + i % no;
    float v = b ? b[oo] : 0.f;
    for (int q = 0; q < KP; ++q) v += part[(q * R + r) * NOUT + oo];
    out[r * ldo + oo] = v;
  }
}

// Barrier over the G workgroups of one scene (bar[0] arrivals, bar[1] generation: the pattern of il_tree_sync, ilqr_kernels.hip).  A workgroup
// that waits ~seconds for its peers (they can only be missing if the launch is not resident) raises the launch's abort word -- host-visible
// memory -- and every workgroup leaves at its next barrier: the host reports the failure instead of a hung device.
#define DEC_SYNC_SPINS (1u << 22)
__device__ __forceinline__ bool dec_grid_sync(unsigned *bar, int G, unsigned *abort_word) {
  __shared__ int sh_ab;
  __syncthreads();
  if (threadIdx.x == 0) {
    int aborted = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned prev = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (unsigned)G - 1u) {
      __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned spins = 0;
      while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 0xfffu) == 0u) {
          if (spins >= DEC_SYNC_SPINS) __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { aborted = 1; break; }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    sh_ab = aborted;
  }
  __syncthreads();
  return sh_ab != 0;
}

// the stage's rows meet: every workgroup publishes its range of out[R][NOUT] (LDS) in `xb` (global), waits for the others and reads the rest
template <int R>
__device__ __forceinline__ bool dec_exchange(float *out, int ldo, int NOUT, int wg, int G, float *xb, unsigned *bar, unsigned *abort_word) {
  const int NQ = NOUT >> 2;
  const int o_lo = (NQ * wg / G) * 4, o_hi = (NQ * (wg + 1) / G) * 4, no = o_hi - o_lo;
  __syncthreads();
  for (int i = threadIdx.x; i < R * no; i += blockDim.x) { const int r = i / no, oo = o_lo + i % no; xb[r * NOUT + oo] = out[r * ldo + oo]; }
  if (dec_grid_sync(bar, G, abort_word)) return true;
  // (all of a thread's loads in flight before its first LDS write: the rows arrive in one L2 round trip instead of one per 1 024 floats)
  constexpr int NV = (R * 1536 / 4 + DT - 1) / DT;          // f32x4 loads per thread at the widest stage
  const int n4 = R * NOUT / 4;
  f32x4 v[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) { const int i = threadIdx.x + q * DT; if (i < n4) v[q] = *(const f32x4 *)(xb + 4 * (size_t)i); }
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int i = threadIdx.x + q * DT;
    if (i < n4) { const int r = (4 * i) / NOUT, oo = (4 * i) % NOUT; if (oo < o_lo || oo >= o_hi) *(f32x4 *)(out + r * ldo + oo) = v[q]; }
  }
  return false;
}

template <int R>
__device__ __forceinline__ void dense(const float *in, int ldi, int K, const float *__restrict__ WT,
                                      const float *__restrict__ b, int NOUT, float *out, int ldo,
                                      float *part = nullptr, int part_floats = 0) {
  const int nthr = blockDim.x;
  if (R <= 8 && part && (NOUT & 3) == 0) {
    dense_quads<R>(in, ldi, K, WT, b, NOUT, out, ldo, part, part_floats);
    return;
  }
  int KP = 1;
  if (part && NOUT < nthr) {
    KP = nthr / NOUT;
    while (KP > 1 && (KP * R * NOUT > part_floats || (K / 4) < KP)) KP >>= 1;
    // power of two
    int p2 = 1;
    while (p2 * 2 <= KP) p2 *= 2;
    KP = p2;
  }
  if (KP == 1) {
    for (int o = threadIdx.x; o < NOUT; o += nthr) {
      float acc[R];
      const float bv = b ? b[o] : 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = bv;
      int k = 0;
      for (; k + DB <= K; k += DB) {
        float w[DB];
#pragma unroll
        for (int q = 0; q < DB; ++q) w[q] = WT[(size_t)(k + q) * NOUT + o];
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
          for (int q4 = 0; q4 < DB / 4; ++q4) {
            const f32x4 x = *(const f32x4 *)(in + r * ldi + k + 4 * q4);
            acc[r] = fmaf(w[4 * q4 + 0], x[0], acc[r]);
            acc[r] = fmaf(w[4 * q4 + 1], x[1], acc[r]);
            acc[r] = fmaf(w[4 * q4 + 2], x[2], acc[r]);
            acc[r] = fmaf(w[4 * q4 + 3], x[3], acc[r]);
          }
        DENSE_ROW_FENCE();
        }
      }
      for (; k < K; k += 4) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const f32x4 x = *(const f32x4 *)(in + r * ldi + k);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[r] = fmaf(WT[(size_t)(k + q) * NOUT + o], x[q], acc[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) out[r * ldo + o] = acc[r];
    }
    return;
  }
  const int kp = threadIdx.x / NOUT, o = threadIdx.x % NOUT;
  const int Kc = ((K / 4 + KP - 1) / KP) * 4;
  if (kp < KP) {
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    const int k0 = kp * Kc, k1 = min(K, k0 + Kc);
    int k = k0;
    for (; k + DB <= k1; k += DB) {
      float w[DB];
#pragma unroll
      for (int q = 0; q < DB; ++q) w[q] = WT[(size_t)(k + q) * NOUT + o];
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int q4 = 0; q4 < DB / 4; ++q4) {
          const f32x4 x = *(const f32x4 *)(in + r * ldi + k + 4 * q4);
          acc[r] = fmaf(w[4 * q4 + 0], x[0], acc[r]);
          acc[r] = fmaf(w[4 * q4 + 1], x[1], acc[r]);
          acc[r] = fmaf(w[4 * q4 + 2], x[2], acc[r]);
          acc[r] = fmaf(w[4 * q4 + 3], x[3], acc[r]);
        }
        DENSE_ROW_FENCE();
      }
    }
    for (; k < k1; k += 4) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const f32x4 x = *(const f32x4 *)(in + r * ldi + k);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[r] = fmaf(WT[(size_t)(k + q) * NOUT + o], x[q], acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) part[(kp * R + r) * NOUT + o] = acc[r];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < R * NOUT; i += nthr) {
    const int r = i / NOUT, oo = i % NOUT;
    float v = b ? b[oo] : 0.f;
    for (int q = 0; q < KP; ++q) v += part[(q * R + r) * NOUT + oo];
    out[r * ldo + oo] = v;
  }
}

// LayerNorm (+ReLU) over `n` features of each of R rows, in place; one wave per row (4 waves).
#define LN_REGS 12      // elements per lane held in registers: rows of up to 768 features
__device__ __forceinline__ void ln_rows(float *buf, int ld, int R, int n, const float *__restrict__ g,
                                        const float *__restrict__ be, bool relu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r = wave; r < R; r += nw) {
    if (n <= 64 * LN_REGS) {
      // the lane's elements and their affine parameters in registers: gamma / beta are requested BEFORE the two reductions instead of
      // one dependent L2 round trip per element behind them (k_dec_scene's stage trace: 10 k cycles for the 768-wide row, 4 k for a
      // 128-wide one).  Same accumulation order as the loop form below: same bits.
      float xv[LN_REGS], gv[LN_REGS], bv[LN_REGS];
#pragma unroll
      for (int i = 0; i < LN_REGS; ++i) {
        const int k = lane + 64 * i;
        const bool in = k < n;
        xv[i] = in ? buf[r * ld + k] : 0.f;
        gv[i] = in ? g[k] : 0.f;
        bv[i] = in ? be[k] : 0.f;
      }
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < LN_REGS; ++i) if (lane + 64 * i < n) s += xv[i];
      const float mean = wave_sum(s) / (float)n;
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < LN_REGS; ++i) if (lane + 64 * i < n) { const float d = xv[i] - mean; v = fmaf(d, d, v); }
      const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)n + 1e-5f);
#pragma unroll
      for (int i = 0; i < LN_REGS; ++i) {
        const int k = lane + 64 * i;
        if (k < n) {
          float y = (xv[i] - mean) * rstd * gv[i] + bv[i];
          if (relu) y = fmaxf(y, 0.f);
          buf[r * ld + k] = y;
        }
      }
      continue;
    }
    float s = 0.f;
    for (int k = lane; k < n; k += 64) s += buf[r * ld + k];
    const float mean = wave_sum(s) / (float)n;
    float v = 0.f;
    for (int k = lane; k < n; k += 64) { const float d = buf[r * ld + k] - mean; v = fmaf(d, d, v); }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)n + 1e-5f);
    for (int k = lane; k < n; k += 64) {
      float y = (buf[r * ld + k] - mean) * rstd * g[k] + be[k];
      if (relu) y = fmaxf(y, 0.f);
      buf[r * ld + k] = y;
    }
  }
}

// =================================================================================================
// LaneNet: PL polylines (10 points each) per workgroup.
// =================================================================================================
struct LaneW {
  const float *pW, *pb, *pg, *pbe;  // proj 16->128 (+LN)
  // aggre blocks 0/1: fc1.0, fc1.3, fc2.0 (256->128), fc2.3, norm
  const float *f10W[2], *f10b[2], *f10g[2], *f10be[2];
  const float *f13W[2], *f13b[2], *f13g[2], *f13be[2];
  const float *f20W[2], *f20b[2], *f20g[2], *f20be[2];
  const float *f23W[2], *f23b[2], *f23g[2], *f23be[2];
  const float *ng[2], *nbe[2];
};

#define PL 2
#define LR (PL * 10)

#ifndef DT
#define DT 1024   // threads of the latency-bound dense kernels (K-split over thread groups)
#endif
__global__ __launch_bounds__(DT) void k_lane_net(const float *__restrict__ feats, int n_poly,
                                                 float *__restrict__ out, LaneW W) {
  __shared__ __attribute__((aligned(16))) float xin[LR][16];
  __shared__ __attribute__((aligned(16))) float x[LR][128];
  __shared__ __attribute__((aligned(16))) float hcat[LR][256];
  __shared__ __attribute__((aligned(16))) float t1[LR][128];
  __shared__ __attribute__((aligned(16))) float part[2 * LR * 128];
  const int PF = 2 * LR * 128;
  const int tid = threadIdx.x;
  const int p0 = blockIdx.x * PL;
  for (int i = tid; i < LR * 16; i += blockDim.x) {
    const int r = i / 16, c = i % 16;
    const int pl = p0 + r / 10;
    xin[r][c] = pl < n_poly ? feats[((size_t)pl * 10 + r % 10) * 16 + c] : 0.f;
  }
  __syncthreads();
  dense<LR>(&xin[0][0], 16, 16, W.pW, W.pb, 128, &x[0][0], 128, part, PF);
  __syncthreads();
  ln_rows(&x[0][0], 128, LR, 128, W.pg, W.pbe, true);
  __syncthreads();
  for (int blk = 0; blk < 2; ++blk) {
    dense<LR>(&x[0][0], 128, 128, W.f10W[blk], W.f10b[blk], 128, &t1[0][0], 128, part, PF);
    __syncthreads();
    ln_rows(&t1[0][0], 128, LR, 128, W.f10g[blk], W.f10be[blk], true);
    __syncthreads();
    dense<LR>(&t1[0][0], 128, 128, W.f13W[blk], W.f13b[blk], 128, &hcat[0][0], 256, part, PF);
    __syncthreads();
    ln_rows(&hcat[0][0], 256, LR, 128, W.f13g[blk], W.f13be[blk], true);
    __syncthreads();
    // max-pool over the 10 points, broadcast into the second half of the concat
    for (int i = tid; i < PL * 128; i += blockDim.x) {
      const int pl = i / 128, c = i % 128;
      float m = hcat[pl * 10][c];
      for (int q = 1; q < 10; ++q) m = fmaxf(m, hcat[pl * 10 + q][c]);
      for (int q = 0; q < 10; ++q) hcat[pl * 10 + q][128 + c] = m;
    }
    __syncthreads();
    dense<LR>(&hcat[0][0], 256, 256, W.f20W[blk], W.f20b[blk], 128, &t1[0][0], 128, part, PF);
    __syncthreads();
    ln_rows(&t1[0][0], 128, LR, 128, W.f20g[blk], W.f20be[blk], true);
    __syncthreads();
    dense<LR>(&t1[0][0], 128, 128, W.f23W[blk], W.f23b[blk], 128, &hcat[0][0], 256, part, PF);
    __syncthreads();
    ln_rows(&hcat[0][0], 256, LR, 128, W.f23g[blk], W.f23be[blk], true);
    __syncthreads();
    for (int i = tid; i < LR * 128; i += blockDim.x) x[i / 128][i % 128] += hcat[i / 128][i % 128];
    __syncthreads();
    ln_rows(&x[0][0], 128, LR, 128, W.ng[blk], W.nbe[blk], false);
    __syncthreads();
  }
  for (int i = tid; i < PL * 128; i += blockDim.x) {
    const int pl = i / 128, c = i % 128;
    if (p0 + pl < n_poly) {
      float m = x[pl * 10][c];
      for (int q = 1; q < 10; ++q) m = fmaxf(m, x[pl * 10 + q][c]);
      out[(size_t)(p0 + pl) * 128 + c] = m;
    }
  }
}

// =================================================================================================
// ActorNet: one agent per workgroup; the whole 4-scale pyramid lives in LDS.
// Conv weights are stored [ci][dk][co] (transposed) so that thread <-> co reads are coalesced.
// =================================================================================================
struct ResW { const float *c1, *g1, *b1, *c2, *g2, *b2, *ds, *gd, *bd; };
struct ActorW {
  ResW res[9];                 // groups.{0..3}.{0,1}, output
  const float *latW[4], *latG[4], *latB[4];
};

#ifndef AT
#define AT 1024   // threads of k_actor_net
#endif

// out[co][t] = sum_ci sum_dk W[ci][dk][co] * in[ci][t*stride + dk - pad], raw (no norm).
// thread = (4 consecutive output channels, time chunk of TCH outputs, K part).  The layer is latency-bound on the
// weight stream, so every load is made as wide as the layout allows (one 16-byte load = the same tap of four
// output channels) and the Cin range is split over as many K parts as the LDS scratch `part` holds
// (KP*Cout*Tout floats; partials reduced there).  Per input channel a thread reads the window of
// (TCH-1)*STRIDE + KSZ inputs it needs ONCE into registers and reuses it for all taps and all four channels.
// Contains __syncthreads().
template <int TCH, int STRIDE, int KSZ>
__device__ __forceinline__ void conv_chunk(const float *in, int Cin, int Tin, const float *__restrict__ W,
                                           int Cout, int Tout, float *out, float *part, int KP) {
  constexpr int SPAN = (TCH - 1) * STRIDE + KSZ, PAD = (KSZ - 1) / 2;
  const int tid = threadIdx.x;
  const int nch = (Tout + TCH - 1) / TCH;
  const int CG = Cout >> 2;                      // channel quads
  const int groups = CG * nch;
  const int kp = tid / groups, g = tid % groups;
  const int co = (g % CG) * 4, t0 = (g / CG) * TCH;
  const int cpk = (Cin + KP - 1) / KP;
  if (kp < KP) {
    float acc[4][TCH];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < TCH; ++i) acc[c][i] = 0.f;
    const int c0 = kp * cpk, c1 = min(Cin, c0 + cpk);
    const int base = t0 * STRIDE - PAD;
#ifndef CONV_UNROLL
#define CONV_UNROLL 4
#endif
#pragma unroll CONV_UNROLL
    for (int ci = c0; ci < c1; ++ci) {
      f32x4 w[KSZ];
      float xw[SPAN];
#pragma unroll
      for (int dk = 0; dk < KSZ; ++dk) w[dk] = *(const f32x4 *)(W + ((size_t)ci * KSZ + dk) * Cout + co);
#pragma unroll
      for (int j = 0; j < SPAN; ++j) {
        const int ti = base + j;
        xw[j] = (ti >= 0 && ti < Tin) ? in[ci * Tin + ti] : 0.f;
      }
#pragma unroll
      for (int dk = 0; dk < KSZ; ++dk)        // tap-major inside a channel
#pragma unroll
        for (int i = 0; i < TCH; ++i) {
          const float x = xw[i * STRIDE + dk];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c][i] = fmaf(w[dk][c], x, acc[c][i]);
        }
      DENSE_ROW_FENCE();      // keep one channel's window next to its FMAs (see dense())
    }
    float *dst = KP > 1 ? part + (size_t)kp * Cout * Tout : out;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < TCH; ++i)
        if (t0 + i < Tout) dst[(co + c) * Tout + t0 + i] = acc[c][i];
  }
  if (KP > 1) {
    __syncthreads();
    for (int i = tid; i < Cout * Tout; i += AT) {
      float v = part[i];
      for (int q = 1; q < KP; ++q) v += part[(size_t)q * Cout * Tout + i];
      out[i] = v;
    }
  }
}

// K split: as many parts as threads and the `part` scratch allow (every layer of the actor net has
// Cout/4 x ceil(Tout/3) <= 512 thread groups; longer time chunks were measured slower: 274 / 289 / 504 us per
// launch for 3 / 6 / 12 outputs per thread on the T = 48 layers).  STRIDE / KSZ are compile-time at every call site.
template <int STRIDE, int KSZ>
__device__ __forceinline__ void conv(const float *in, int Cin, int Tin, const float *W, int Cout, int Tout,
                                     float *out, float *part, int part_floats) {
const int groups = (Cout >> 2) * ((Tout + 2) / 3);
  int KP = AT / groups;
  while (KP > 1 && (KP * Cout * Tout > part_floats || KP > Cin)) KP >>= 1;
  if (KP < 1) KP = 1;
  conv_chunk<3, STRIDE, KSZ>(in, Cin, Tin, W, Cout, Tout, out, part, KP);
}

// sum over the 16 waves of the actor kernel.  `red` holds two 16-float buffers used alternately (`flip`), so one
// barrier per reduction is enough: a buffer is rewritten only after the barrier of the reduction in between.
__device__ __forceinline__ float block_sum_a(float v, float *red, int flip) {
  v = wave_sum(v);
  float *r = red + (flip & 1) * 16;
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < AT / 64; ++w) s += r[w];
  return s;
}

// GroupNorm(1 group) over C x T (<= 6 * AT elements), per-channel affine, optional residual add and ReLU, in place.
// Every thread keeps its elements in registers across the three phases and requests its affine parameters (global
// memory) before the first reduction, so their latency hides behind the two block reductions.
#define GN_NE 6
__device__ __forceinline__ void gn(float *buf, int C, int T, const float *__restrict__ g,
                                   const float *__restrict__ b, const float *resid, bool relu, float *red) {
  const int n = C * T;
  float x[GN_NE], gg[GN_NE], bb[GN_NE];
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < GN_NE; ++e) {
    const int i = threadIdx.x + e * AT;
    if (i < n) {
      const int c = i / T;
      x[e] = buf[i];
      gg[e] = g[c];
      bb[e] = b[c];
      s += x[e];
    }
  }
  const float mean = block_sum_a(s, red, 0) / (float)n;
  float v = 0.f;
#pragma unroll
  for (int e = 0; e < GN_NE; ++e)
    if ((int)threadIdx.x + e * AT < n) { const float d = x[e] - mean; v = fmaf(d, d, v); }
  const float rstd = 1.0f / sqrtf(block_sum_a(v, red, 1) / (float)n + 1e-5f);
#pragma unroll
  for (int e = 0; e < GN_NE; ++e) {
    const int i = threadIdx.x + e * AT;
    if (i < n) {
      float y = (x[e] - mean) * rstd * gg[e] + bb[e];
      if (resid) y += resid[i];
      if (relu) y = fmaxf(y, 0.f);
      buf[i] = y;
    }
  }
  __syncthreads();
}

// Res1d (layers.py:175-188): out written to `out`; uses scratch t1, t2 (each >= Cout*Tout floats)
__device__ __forceinline__ void res1d(const float *in, int Cin, int Tin, const ResW &w, int Cout, int stride,
                                      float *out, float *t1, float *t2, float *red, float *part, int pf) {
  const int Tout = Tin / stride;
  if (stride == 1) conv<1, 3>(in, Cin, Tin, w.c1, Cout, Tout, t1, part, pf);
  else conv<2, 3>(in, Cin, Tin, w.c1, Cout, Tout, t1, part, pf);
  __syncthreads();
  gn(t1, Cout, Tout, w.g1, w.b1, nullptr, true, red);
  conv<1, 3>(t1, Cout, Tout, w.c2, Cout, Tout, out, part, pf);
  __syncthreads();
  const float *resid = in;
  if (w.ds) {
    if (stride == 1) conv<1, 1>(in, Cin, Tin, w.ds, Cout, Tout, t2, part, pf);
    else conv<2, 1>(in, Cin, Tin, w.ds, Cout, Tout, t2, part, pf);
    __syncthreads();
      gn(t2, Cout, Tout, w.gd, w.bd, nullptr, false, red);
      resid = t2;
  }
  gn(out, Cout, Tout, w.g2, w.b2, resid, true, red);
}

// LDS carve (floats): xin 672 | o0 1536 | o1 1536 | o2 1536 | o3 1536 | ta 1536 | tb 1536 | tc 1536
//                     | fa 6144 | fb 6144 ; the final Res1d output reuses the (dead) o0..tc region.
#define ACT_PART 12288
#define ACT_LDS_FLOATS (672 + 7 * 1536 + 2 * 6144 + 32 + ACT_PART)
__global__ __launch_bounds__(AT) void k_actor_net(const float *__restrict__ actors, int n_actors,
                                                  float *__restrict__ out, ActorW W) {
  extern __shared__ float sm[];
  float *xin = sm;
  float *o0 = sm + 672, *o1 = o0 + 1536, *o2 = o1 + 1536, *o3 = o2 + 1536;
  float *ta = o3 + 1536, *tb = ta + 1536, *tc = tb + 1536;
  float *fa = tc + 1536, *fb = fa + 6144;
  float *red = fb + 6144;
  float *part = red + 32;
  const int pf = ACT_PART;
  float *fo = o0;  // 10752 floats available, needs 6144
  const int tid = threadIdx.x;
  const int a = blockIdx.x;
  if (a >= n_actors) return;
#ifdef MIND_ENC_TRACE
  long long at_[12], aq_ = clock64();
  int an_ = 0;
#define AT_MARK() do { const long long n_ = clock64(); at_[an_++] = n_ - aq_; aq_ = n_; } while (0)
#else
#define AT_MARK() do {} while (0)
#endif
  for (int i = tid; i < 14 * 48; i += AT) xin[i] = actors[(size_t)a * 14 * 48 + i];
  __syncthreads();
  AT_MARK();
  res1d(xin, 14, 48, W.res[0], 32, 1, ta, tb, tc, red, part, pf);
  res1d(ta, 32, 48, W.res[1], 32, 1, o0, tb, tc, red, part, pf);
  AT_MARK();
  res1d(o0, 32, 48, W.res[2], 64, 2, ta, tb, tc, red, part, pf);
  res1d(ta, 64, 24, W.res[3], 64, 1, o1, tb, tc, red, part, pf);
  AT_MARK();
  res1d(o1, 64, 24, W.res[4], 128, 2, ta, tb, tc, red, part, pf);
  res1d(ta, 128, 12, W.res[5], 128, 1, o2, tb, tc, red, part, pf);
  AT_MARK();
  res1d(o2, 128, 12, W.res[6], 256, 2, ta, tb, tc, red, part, pf);
  res1d(ta, 256, 6, W.res[7], 256, 1, o3, tb, tc, red, part, pf);
  AT_MARK();
  // FPN top-down (network.py:55-58): lateral = conv3 + GN, no activation
  conv<1, 3>(o3, 256, 6, W.latW[3], 128, 6, fa, part, pf);
  __syncthreads();
  gn(fa, 128, 6, W.latG[3], W.latB[3], nullptr, false, red);
  float *cur = fa, *nxt = fb;
  for (int g = 2; g >= 0; --g) {
    const float *src_o = g == 2 ? o2 : (g == 1 ? o1 : o0);
    const int C = g == 2 ? 128 : (g == 1 ? 64 : 32);
    const int T = g == 2 ? 12 : (g == 1 ? 24 : 48);
    const int Th = T / 2;
    conv<1, 3>(src_o, C, T, W.latW[g], 128, T, nxt, part, pf);
    __syncthreads();
    gn(nxt, 128, T, W.latG[g], W.latB[g], nullptr, false, red);
    // x2 linear upsample of `cur` (align_corners=False) added to the lateral
    for (int i = tid; i < 128 * T; i += AT) {
      const int c = i / T, t = i % T;
      float src = (t + 0.5f) * 0.5f - 0.5f;
      src = src < 0.f ? 0.f : src;
      const int i0 = (int)src;
      const int i1 = i0 + 1 < Th ? i0 + 1 : Th - 1;
      const float l1 = src - (float)i0;
      nxt[i] += (1.0f - l1) * cur[c * Th + i0] + l1 * cur[c * Th + i1];
    }
    __syncthreads();
    float *t = cur; cur = nxt; nxt = t;
  }
  AT_MARK();
  // output Res1d(128,128) at T = 48 (network.py:60); only the last time column is kept
  {
    const ResW &w = W.res[8];
    conv<1, 3>(cur, 128, 48, w.c1, 128, 48, nxt, part, pf);
    __syncthreads();
    gn(nxt, 128, 48, w.g1, w.b1, nullptr, true, red);
    conv<1, 3>(nxt, 128, 48, w.c2, 128, 48, fo, part, pf);
    __syncthreads();
    gn(fo, 128, 48, w.g2, w.b2, cur, true, red);
  }
  AT_MARK();
  if (tid < 128) out[(size_t)a * 128 + tid] = fo[tid * 48 + 47];
#ifdef MIND_ENC_TRACE
  if (a == 0 && tid == 0)
    printf("[k_actor_net n=%d] cycles: load %lld group0 %lld group1 %lld group2 %lld group3 %lld fpn %lld out-res %lld\n", n_actors, at_[0], at_[1], at_[2],
           at_[3], at_[4], at_[5], at_[6]);
#endif
}
extern "C" size_t mind_actor_lds_bytes() { return (size_t)ACT_LDS_FLOATS * sizeof(float); }

// =================================================================================================
// SceneDecoder, scene part: target embedding (k_dec_tgt), 6 mode tokens through ctx_proj + 2 encoder layers,
// mode probabilities (k_dec_scene).  One workgroup per scene.
// =================================================================================================
struct DecW {
  const float *rpeW, *rpeb, *rpeg, *rpebe;                       // proj_rpe 20(->pad 20)->128
  const float *t0W, *t0b, *t0g, *t0be, *t3W, *t3b, *t3g, *t3be;  // proj_tgt
  const float *c0W, *c0b, *c0g, *c0be, *c3W, *c3b, *c3g, *c3be;  // ctx_proj 128->384->768
  const float *a0W, *a0b, *a0g, *a0be, *a3W, *a3b, *a3g, *a3be;  // actor_proj
  const float *inW[2], *inb[2], *outW[2], *outb[2], *l1W[2], *l1b[2], *l2W[2], *l2b[2];
  const float *n1g[2], *n1b[2], *n2g[2], *n2b[2];
  const float *k0W, *k0b, *k0g, *k0be, *k3W, *k3b, *k3g, *k3be, *k6W, *k6b;  // cls head
  const float *r0W, *r0b, *r0g, *r0be, *r3W, *r3b, *r3g, *r3be, *r6W, *r6b;  // reg head
  const float *T, *Tp;                                                           // [60][8], [60][7]
};

#define DEC_SCENE_PART 24576
#define DEC_SCENE_LDS_FLOATS (256 + 768 + 768 + 2304 + 768 + 9216 + 768 + 144 + DEC_SCENE_PART)
// target embedding (network.py:491-495): needs only the target polyline's LaneNet feature and its RPE, so it runs on the side stream
// right behind that LaneNet launch, off the critical path of the fusion layers.  One workgroup per scene.
__global__ __launch_bounds__(DT) void k_dec_tgt(const float *__restrict__ tgt_feat /*[B,128]*/, const float *__restrict__ tgt_rpe /*[B,20]*/,
                                                float *__restrict__ tgt_out /*[B,128]*/, DecW W) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  float (*v0)[256] = (float (*)[256])(dsm);
  float (*v1)[768] = (float (*)[768])(dsm + 256);
  float *part = dsm + 14992;
  const int PF = DEC_SCENE_PART;
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  if (tid < 20) v1[0][tid] = tgt_rpe[(size_t)b * 20 + tid];
  __syncthreads();
  dense<1>(&v1[0][0], 768, 20, W.rpeW, W.rpeb, 128, &v0[0][128], 256, part, PF);
  if (tid < 128) v0[0][tid] = tgt_feat[(size_t)b * 128 + tid];
  __syncthreads();
  ln_rows(&v0[0][128], 256, 1, 128, W.rpeg, W.rpebe, true);
  __syncthreads();
  dense<1>(&v0[0][0], 256, 256, W.t0W, W.t0b, 128, &v1[0][0], 768, part, PF);
  __syncthreads();
  ln_rows(&v1[0][0], 768, 1, 128, W.t0g, W.t0be, true);
  __syncthreads();
  dense<1>(&v1[0][0], 768, 128, W.t3W, W.t3b, 128, &v0[0][0], 256, part, PF);
  __syncthreads();
  ln_rows(&v0[0][0], 256, 1, 128, W.t3g, W.t3be, true);
  __syncthreads();
  if (tid < 128) tgt_out[(size_t)b * 128 + tid] = v0[0][tid];
}

#ifdef MIND_DEC_TRACE
__device__ long long dec_tr_[64];
__device__ int dec_trn_;
#define DEC_MARK() do { if (blockIdx.x == 0 && threadIdx.x == 0) { dec_tr_[dec_trn_ & 63] = clock64(); ++dec_trn_; } } while (0)
#else
#define DEC_MARK() do {} while (0)
#endif
// MW (k_dec_scene_mw): G workgroups per scene.  Every workgroup runs the whole decoder -- the small stages redundantly, which costs no time --
// except the five stages that stream 0.8-1.2 MB of weights through one CU's L2 port (ctx_proj's second layer, the two feed-forward layers of
// both encoder layers: 57 % of the one-workgroup kernel's time at ~60 % of the port's rate): there a workgroup computes its eighth of the
// outputs (dense_quads_part: the same K split and per-item arithmetic, the same bits) and the rows meet through global memory and a
// barrier of the scene's workgroups (dec_exchange; two alternating buffers: a workgroup can be at most one stage ahead of the slowest).
// Block b -> scene 8 (b / 8G) + b % 8, workgroup (b % 8G) / 8: a scene's workgroups share an XCD (one L2), as in k_ilqr.
// PART 0: the whole scene part; 1: up to the mode tokens C (what the actor part needs); 2: the cls head alone, from the C of a PART 1 launch
// (k_dec_cls: on the side stream beside the actor part's head -- the same code on the same values: the same bits)
template <bool MW, int PART = 0>
__device__ __forceinline__ void k_dec_scene_body(const float *__restrict__ x /*[tokens,128]*/,
                                                 const int *__restrict__ cls_row /*[B]*/,
                                                 float *__restrict__ Cout /*[B,6,128]*/,
                                                 float *__restrict__ cls_out /*[B,6]*/, const DecW &W, int b, int wg, int G,
                                                 float *xbuf, unsigned *bar, unsigned *abort_word) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  float (*v0)[256] = (float (*)[256])(dsm);
  float (*v1)[768] = (float (*)[768])(dsm + 256);
  float (*C)[128] = (float (*)[128])(dsm + 1024);
  float (*qkv)[384] = (float (*)[384])(dsm + 1792);
  float (*att)[128] = (float (*)[128])(dsm + 4096);
  float (*ff)[1536] = (float (*)[1536])(dsm + 4864);
  float (*t2)[128] = (float (*)[128])(dsm + 14080);
  float (*sc)[6][6] = (float (*)[6][6])(dsm + 14848);
  float *part = dsm + 14992;
  const int PF = DEC_SCENE_PART;
  const int tid = threadIdx.x;
  int xstage = 0;                           // big stages so far (MW): selects the exchange buffer
  float *xb0 = xbuf, *xb1 = xbuf + 6 * 1536;
#ifdef MIND_DEC_TRACE
  if (blockIdx.x == 0 && tid == 0) dec_trn_ = 0;
#endif
  DEC_MARK();
  if (PART == 2) {
    for (int i = tid; i < 768; i += blockDim.x) C[i / 128][i % 128] = Cout[(size_t)b * 768 + i];
    __syncthreads();
  }
  if (PART != 2) {
  // ---- ctx_proj: cls token -> 6 mode tokens (network.py:501)
  if (tid < 128) v0[0][tid] = x[(size_t)cls_row[b] * 128 + tid];
  __syncthreads();
  dense<1>(&v0[0][0], 256, 128, W.c0W, W.c0b, 384, &v1[0][0], 768, part, PF);
  DEC_MARK();
  __syncthreads();
  ln_rows(&v1[0][0], 768, 1, 384, W.c0g, W.c0be, true);
  DEC_MARK();
  __syncthreads();
  if (MW) {
    dense_quads_part<1>(&v1[0][0], 768, 384, W.c3W, W.c3b, 768, &ff[0][0], 1536, part, PF, wg, G);
    if (dec_exchange<1>(&ff[0][0], 1536, 768, wg, G, (xstage++ & 1) ? xb1 : xb0, bar, abort_word)) return;
  } else
    dense<1>(&v1[0][0], 768, 384, W.c3W, W.c3b, 768, &ff[0][0], 1536, part, PF);
  DEC_MARK();
  __syncthreads();
  ln_rows(&ff[0][0], 1536, 1, 768, W.c3g, W.c3be, true);
  DEC_MARK();
  __syncthreads();
  for (int i = tid; i < 768; i += blockDim.x) C[i / 128][i % 128] = ff[0][i];
  __syncthreads();
  // ---- 2 post-norm encoder layers over the 6 mode tokens (4 heads x 32, ffn 1536)
  for (int L = 0; L < 2; ++L) {
    dense<6>(&C[0][0], 128, 128, W.inW[L], W.inb[L], 384, &qkv[0][0], 384, part, PF);
  DEC_MARK();
    __syncthreads();
    if (tid < 4 * 36) {
      const int hd = tid / 36, s = (tid % 36) / 6, t = tid % 6;
      float d = 0.f;
      for (int k = 0; k < 32; ++k) d = fmaf(qkv[s][hd * 32 + k], qkv[t][128 + hd * 32 + k], d);
      sc[hd][s][t] = d / sqrtf(32.0f);
    }
    __syncthreads();
    if (tid < 24) {
      const int hd = tid / 6, s = tid % 6;
      float m = -INFINITY;
      for (int t = 0; t < 6; ++t) m = fmaxf(m, sc[hd][s][t]);
      float e[6], sum = 0.f;
      for (int t = 0; t < 6; ++t) { e[t] = expf(sc[hd][s][t] - m); sum += e[t]; }
      for (int t = 0; t < 6; ++t) sc[hd][s][t] = e[t] / sum;
    }
    __syncthreads();
    for (int i = tid; i < 6 * 128; i += blockDim.x) {
      const int s = i / 128, c = i % 128, hd = c / 32;
      float o = 0.f;
      for (int t = 0; t < 6; ++t) o = fmaf(sc[hd][s][t], qkv[t][256 + c], o);
      att[s][c] = o;
    }
    __syncthreads();
    dense<6>(&att[0][0], 128, 128, W.outW[L], W.outb[L], 128, &t2[0][0], 128, part, PF);
  DEC_MARK();
    __syncthreads();
    for (int i = tid; i < 768; i += blockDim.x) C[i / 128][i % 128] += t2[i / 128][i % 128];
    __syncthreads();
    ln_rows(&C[0][0], 128, 6, 128, W.n1g[L], W.n1b[L], false);
  DEC_MARK();
    __syncthreads();
    if (MW) {
      dense_quads_part<6>(&C[0][0], 128, 128, W.l1W[L], W.l1b[L], 1536, &ff[0][0], 1536, part, PF, wg, G);
      if (dec_exchange<6>(&ff[0][0], 1536, 1536, wg, G, (xstage++ & 1) ? xb1 : xb0, bar, abort_word)) return;
    } else
      dense<6>(&C[0][0], 128, 128, W.l1W[L], W.l1b[L], 1536, &ff[0][0], 1536, part, PF);
  DEC_MARK();
    __syncthreads();
    for (int i = tid; i < 6 * 1536; i += blockDim.x) ff[i / 1536][i % 1536] = fmaxf(ff[i / 1536][i % 1536], 0.f);
    __syncthreads();
    if (MW) {
      dense_quads_part<6>(&ff[0][0], 1536, 1536, W.l2W[L], W.l2b[L], 128, &t2[0][0], 128, part, PF, wg, G);
      if (dec_exchange<6>(&t2[0][0], 128, 128, wg, G, (xstage++ & 1) ? xb1 : xb0, bar, abort_word)) return;
    } else
      dense<6>(&ff[0][0], 1536, 1536, W.l2W[L], W.l2b[L], 128, &t2[0][0], 128, part, PF);
  DEC_MARK();
    __syncthreads();
    for (int i = tid; i < 768; i += blockDim.x) C[i / 128][i % 128] += t2[i / 128][i % 128];
    __syncthreads();
    ln_rows(&C[0][0], 128, 6, 128, W.n2g[L], W.n2b[L], false);
  DEC_MARK();
    __syncthreads();
  }
  if (!MW || wg == 0)
    for (int i = tid; i < 768; i += blockDim.x) Cout[(size_t)b * 768 + i] = C[i / 128][i % 128];
  }
  if (PART == 1) return;
  // ---- cls head on the mode tokens only (network.py:512, Q6), softmax over the 6 modes
  dense<6>(&C[0][0], 128, 128, W.k0W, W.k0b, 128, &att[0][0], 128, part, PF);
  DEC_MARK();
  __syncthreads();
  ln_rows(&att[0][0], 128, 6, 128, W.k0g, W.k0be, true);
  DEC_MARK();
  __syncthreads();
  dense<6>(&att[0][0], 128, 128, W.k3W, W.k3b, 128, &t2[0][0], 128, part, PF);
  DEC_MARK();
  __syncthreads();
  ln_rows(&t2[0][0], 128, 6, 128, W.k3g, W.k3be, true);
  DEC_MARK();
  __syncthreads();
  for (int k = tid >> 6; k < 6; k += (int)(blockDim.x >> 6)) {
    const int lane = tid & 63;
    float s = t2[k][lane] * W.k6W[lane] + t2[k][lane + 64] * W.k6W[lane + 64];
    s = wave_sum(s);
    if (lane == 0) sc[0][0][k] = s + W.k6b[0];
  }
  __syncthreads();
  if (tid == 0 && (!MW || wg == 0)) {
    float m = -INFINITY;
    for (int k = 0; k < 6; ++k) m = fmaxf(m, sc[0][0][k]);
    float e[6], sum = 0.f;
    for (int k = 0; k < 6; ++k) { e[k] = expf(sc[0][0][k] - m); sum += e[k]; }
    for (int k = 0; k < 6; ++k) cls_out[(size_t)b * 6 + k] = e[k] / sum;
  }
  DEC_MARK();
#ifdef MIND_DEC_TRACE
  if (blockIdx.x == 0 && tid == 0) {
    printf("[k_dec_scene] cycles between marks (dense / LayerNorm stages in program order):");
    for (int q = 1; q < dec_trn_ && q < 64; ++q) printf(" %lld", dec_tr_[q] - dec_tr_[q - 1]);
    printf("\n");
  }
#endif
}

__global__ __launch_bounds__(DT) void k_dec_scene(const float *__restrict__ x, const int *__restrict__ cls_row, float *__restrict__ Cout,
                                                  float *__restrict__ cls_out, DecW W) {
  k_dec_scene_body<false>(x, cls_row, Cout, cls_out, W, (int)blockIdx.x, 0, 1, nullptr, nullptr, nullptr);
}
// the scene part in two launches: the mode tokens (k_dec_scene_c, context stream: the actor part's head waits for nothing else of it) and the
// mode probabilities (k_dec_cls, side stream, beside that head)
__global__ __launch_bounds__(DT) void k_dec_scene_c(const float *__restrict__ x, const int *__restrict__ cls_row, float *__restrict__ Cout, DecW W) {
  k_dec_scene_body<false, 1>(x, cls_row, Cout, nullptr, W, (int)blockIdx.x, 0, 1, nullptr, nullptr, nullptr);
}
__global__ __launch_bounds__(DT) void k_dec_cls(float *__restrict__ Cout, float *__restrict__ cls_out, DecW W) {
  k_dec_scene_body<false, 2>(nullptr, nullptr, Cout, cls_out, W, (int)blockIdx.x, 0, 1, nullptr, nullptr, nullptr);
}
#define DEC_MW_G 8
// xbuf [B][2][6 x 1536] floats, bars [B][4] words (zero once, at allocation: the barrier resets itself), abort_word host-visible
__global__ __launch_bounds__(DT) void k_dec_scene_mw(const float *__restrict__ x, const int *__restrict__ cls_row, float *__restrict__ Cout,
                                                     float *__restrict__ cls_out, DecW W, int n_scenes, float *__restrict__ xbuf,
                                                     unsigned *__restrict__ bars, unsigned *abort_word) {
  const int r = (int)blockIdx.x % (8 * DEC_MW_G);
  const int b = 8 * ((int)blockIdx.x / (8 * DEC_MW_G)) + (r & 7), wg = r >> 3;
  if (b >= n_scenes) return;
  k_dec_scene_body<true>(x, cls_row, Cout, cls_out, W, b, wg, DEC_MW_G, xbuf + (size_t)b * 2 * 6 * 1536, bars + 4 * (size_t)b, abort_word);
}

// =================================================================================================
// SceneDecoder, actor part: RA agents per workgroup -> reg [a,6,60,5], vel [a,6,60,2]
// =================================================================================================
#define RA 4
#define DEC_ACTOR_LDS_FLOATS (512 + 1536 + 3072 + 3072 + 3072 + 960 + 6144)
// PART 0: the whole actor part.  PART 1 / 2: its two halves as separate launches -- actor_proj (128 -> 384 -> 768, needs only the
// fused actor tokens: runs on the side stream beside k_dec_scene) writes h2g [A,768]; the head (mode embedding, reg head, Bezier)
// reads it once k_dec_scene's Cmode / tgt are there.  Same code, same arithmetic per element in all three.
template <int PART>
__global__ __launch_bounds__(DT) void k_dec_actor(const float *__restrict__ x /*[tokens,128]*/,
                                                  const int *__restrict__ actor_row /*[A]*/,
                                                  const int *__restrict__ actor_scene /*[A]*/, int n_actors,
                                                  const float *__restrict__ Cmode /*[B,6,128]*/,
                                                  const float *__restrict__ tgt /*[B,128]*/,
                                                  float *__restrict__ reg, float *__restrict__ vel, DecW W, float *__restrict__ h2g) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  float (*xin)[128] = (float (*)[128])(dsm);
  float (*h1)[384] = (float (*)[384])(dsm + 512);
  float (*h2)[768] = (float (*)[768])(dsm + 2048);
  float (*E)[128] = (float (*)[128])(dsm + 5120);
  float (*t1)[128] = (float (*)[128])(dsm + 8192);
  float (*prm)[40] = (float (*)[40])(dsm + 11264);
  float *part = dsm + 12224;
  const int PF = 6144;
  const int tid = threadIdx.x;
  const int a0 = blockIdx.x * RA;
  if (PART != 2) {
    for (int i = tid; i < RA * 128; i += blockDim.x) {
      const int r = i / 128, a = a0 + r;
      xin[r][i % 128] = a < n_actors ? x[(size_t)actor_row[a] * 128 + i % 128] : 0.f;
    }
    __syncthreads();
    dense<RA>(&xin[0][0], 128, 128, W.a0W, W.a0b, 384, &h1[0][0], 384, part, PF);
    __syncthreads();
    ln_rows(&h1[0][0], 384, RA, 384, W.a0g, W.a0be, true);
    __syncthreads();
    dense<RA>(&h1[0][0], 384, 384, W.a3W, W.a3b, 768, &h2[0][0], 768, part, PF);
    __syncthreads();
    ln_rows(&h2[0][0], 768, RA, 768, W.a3g, W.a3be, true);
    __syncthreads();
    if (PART == 1) {
      for (int i = tid; i < RA * 768; i += blockDim.x) {
        const int r = i / 768, a = a0 + r;
        if (a < n_actors) h2g[(size_t)a * 768 + i % 768] = h2[r][i % 768];
      }
      return;
    }
  } else {
    for (int i = tid; i < RA * 768; i += blockDim.x) {
      const int r = i / 768, a = a0 + r;
      h2[r][i % 768] = a < n_actors ? h2g[(size_t)a * 768 + i % 768] : 0.f;
    }
    __syncthreads();
  }
  // embed = cls_embed + actor_embed (+ tgt on mode 0 only)  (network.py:506-510)
  for (int i = tid; i < RA * 768; i += blockDim.x) {
    const int r = i / 768, k = (i % 768) / 128, c = i % 128;
    const int a = a0 + r;
    float v = 0.f;
    if (a < n_actors) {
      const int sc = actor_scene[a];
      v = h2[r][k * 128 + c] + Cmode[((size_t)sc * 6 + k) * 128 + c];
      if (k == 0) v += tgt[(size_t)sc * 128 + c];
    }
    E[r * 6 + k][c] = v;
  }
  __syncthreads();
  dense<RA * 6>(&E[0][0], 128, 128, W.r0W, W.r0b, 128, &t1[0][0], 128, part, PF);
  __syncthreads();
  ln_rows(&t1[0][0], 128, RA * 6, 128, W.r0g, W.r0be, true);
  __syncthreads();
  dense<RA * 6>(&t1[0][0], 128, 128, W.r3W, W.r3b, 128, &E[0][0], 128, part, PF);
  __syncthreads();
  ln_rows(&E[0][0], 128, RA * 6, 128, W.r3g, W.r3be, true);
  __syncthreads();
  // 128 -> 40 = 8 control points x (x, y, sx, sy, rho)
  for (int i = tid; i < RA * 6 * 40; i += blockDim.x) {
    const int r = i / 40, o = i % 40;
    float acc = W.r6b[o];
    for (int k = 0; k < 128; ++k) acc = fmaf(W.r6W[k * 40 + o], E[r][k], acc);
    prm[r][o] = acc;
  }
  __syncthreads();
  // Bezier evaluation (network.py:515-523,545): pos = T P, cov = exp(T S), vel = Tp dP / 6
  for (int i = tid; i < RA * 6 * 60; i += blockDim.x) {
    const int r = i / 60, t = i % 60;
    const int a = a0 + r / 6, k = r % 6;
    if (a >= n_actors) continue;
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

extern "C" size_t mind_dec_scene_lds_bytes() { return (size_t)DEC_SCENE_LDS_FLOATS * sizeof(float); }
extern "C" size_t mind_dec_actor_lds_bytes() { return (size_t)DEC_ACTOR_LDS_FLOATS * sizeof(float); }
