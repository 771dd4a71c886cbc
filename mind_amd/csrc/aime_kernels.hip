// AIME glue on the device (k7): world-frame conversion of ALL K modes of every agent of a round's scenes,
// heading, max-sigma covariance and the topology signatures that prune_merge compares.
//
// Reference semantics: planners/mind/scenario_tree.py:281-412 (prune_merge) -- per mode
//   traj_pos = ((reg[..., :2] R_i^T) + ctr_i) R^T + orig        (:328-334, R_i = rot(atan2(vec_i)))
//   traj_vel = (vel R_i^T) R^T ;  traj_ang = atan2(vel_y, vel_x) + theta_i + theta_scene   (:336-343)
//   traj_cov = max(sigma_x, sigma_y) + last history covariance   (:345-347, utils.py:536)
//   topology = sum_t wrap(phi_{t+1} - phi_t), phi = angle of (exo - ego) / |exo - ego|  (:361-377)
// float32 like the reference; one 64-lane block per (agent, mode), lane = time step.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct AimeScene {
  int a0, a1;               // agent rows of the scene in the round's batch (a0 = ego)
  int last, pad;            // index of the last predicted step that fits seq_len (may be < 0)
  float r00, r01, r10, r11; // scene rotation ROT (row-major)
  float ox, oy, theta_g, pad2;
};

#define AIME_T 60
#define AIME_K 6
#define AIME_PK 6   // packed world-frame record per (agent, mode, step): x, y, vx, vy, heading, max-sigma

__device__ __forceinline__ void aime_to_world(float lx, float ly, float c, float s, float cx, float cy, const AimeScene &S,
                                              bool translate, float &wx, float &wy) {
  float ax = lx * c + ly * (-s), ay = lx * s + ly * c;
  if (translate) { ax += cx; ay += cy; }
  wx = ax * S.r00 + ay * S.r01;
  wy = ax * S.r10 + ay * S.r11;
  if (translate) { wx += S.ox; wy += S.oy; }
}

__global__ __launch_bounds__(64) void k_aime_world(const AimeScene *__restrict__ scenes, const int *__restrict__ agent_scene,
                                                   const float *__restrict__ reg, const float *__restrict__ vel,
                                                   const float *__restrict__ ctrs, const float *__restrict__ vecs,
                                                   const float *__restrict__ cov_last, float *__restrict__ world,
                                                   float *__restrict__ topo, float *__restrict__ ego_end,
                                                   const float *__restrict__ lane, int n_lane) {
  const int i = blockIdx.x / AIME_K, k = blockIdx.x % AIME_K, t = threadIdx.x;
  const int b = agent_scene[i];
  const AimeScene S = scenes[b];
  const bool live = t < AIME_T;
  const int tc = live ? t : AIME_T - 1;
  const float th = atan2f(vecs[2 * i + 1], vecs[2 * i]);
  const float c = cosf(th), s = sinf(th);
  const size_t e = ((size_t)i * AIME_K + k) * AIME_T + tc;
  const float lx = reg[e * 5], ly = reg[e * 5 + 1], sgx = reg[e * 5 + 2], sgy = reg[e * 5 + 3];
  const float lvx = vel[e * 2], lvy = vel[e * 2 + 1];
  float wx, wy, wvx, wvy;
  aime_to_world(lx, ly, c, s, ctrs[2 * i], ctrs[2 * i + 1], S, true, wx, wy);
  aime_to_world(lvx, lvy, c, s, 0.f, 0.f, S, false, wvx, wvy);
  const float ang = atan2f(lvy, lvx) + th + S.theta_g;
  const float cv = fmaxf(sgx, sgy) + cov_last[i];
  if (live) {
    float *w = world + e * AIME_PK;
    w[0] = wx; w[1] = wy; w[2] = wvx; w[3] = wvy; w[4] = ang; w[5] = cv;
  }
  if (i == S.a0 && S.last >= 0) {
    // ego end point of this mode (step `last`) and its distance to the target lane (min over segments of the
    // distance to the clamped projection, utils.py:486-513), the quantity prune_merge thresholds (:300-306)
    const float px = __shfl(wx, S.last, 64), py = __shfl(wy, S.last, 64), pc = __shfl(cv, S.last, 64);
    float dmin = INFINITY;
    for (int sgi = t; sgi < n_lane - 1; sgi += 64) {
      const float ax = lane[2 * sgi], ay = lane[2 * sgi + 1];
      const float sx = lane[2 * sgi + 2] - ax, sy = lane[2 * sgi + 3] - ay;
      float tt = ((px - ax) * sx + (py - ay) * sy) / (sx * sx + sy * sy);
      tt = fminf(fmaxf(tt, 0.f), 1.f);
      const float dx = (ax + tt * sx) - px, dy = (ay + tt * sy) - py;
      dmin = fminf(dmin, sqrtf(dx * dx + dy * dy));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, o, 64));
    if (t == 0) {
      float *o = ego_end + ((size_t)b * AIME_K + k) * 4;
      o[0] = px; o[1] = py; o[2] = pc; o[3] = dmin;
    }
  }
  // ---- topology signature against the scene's ego (same mode, same step)
  float sig = 0.f;
  if (i != S.a0) {
    const int g = S.a0;
    const float thg = atan2f(vecs[2 * g + 1], vecs[2 * g]);
    const size_t eg = ((size_t)g * AIME_K + k) * AIME_T + tc;
    float ex, ey;
    aime_to_world(reg[eg * 5], reg[eg * 5 + 1], cosf(thg), sinf(thg), ctrs[2 * g], ctrs[2 * g + 1], S, true, ex, ey);
    float rx = wx - ex, ry = wy - ey;
    const float nrm = sqrtf(rx * rx + ry * ry);
    rx /= nrm; ry /= nrm;
    const float phi = atan2f(ry, rx);
    const float phin = __shfl_down(phi, 1, 64);
    float d = phin - phi;
    d = atan2f(sinf(d), cosf(d));
    if (t >= AIME_T - 1) d = 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
    sig = d;
  }
  if (t == 0) topo[(size_t)i * AIME_K + k] = sig;
}
