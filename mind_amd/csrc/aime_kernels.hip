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
  int last, cmp;            // index of the last predicted step that fits seq_len (may be < 0); predicted step the branch-time test
                            // compares against (k_aime_branch: CUR_T, or 1 when CUR_T == 0)
  float r00, r01, r10, r11; // scene rotation ROT (row-major)
  float ox, oy, theta_g, pad2;
};

// The scene tables of a SMALL round by value in the kernel arguments (mind_aime_plan: at most AIME_SMALL scenes per chunk, all with the
// same agent count): the three glue kernels then need no table upload between the predictor and themselves.  n = 0: tables in memory.
#define AIME_SMALL 8
struct AimeSmall {
  AimeScene s[AIME_SMALL];
  float prob[AIME_SMALL];
  int n, a, pad0, pad1;       // scenes, agents per scene (agent row i belongs to scene i / a)
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
                                                   const float *__restrict__ lane, int n_lane, const AimeSmall sm) {
  const int i = blockIdx.x / AIME_K, k = blockIdx.x % AIME_K, t = threadIdx.x;
  const int b = sm.n ? i / sm.a : agent_scene[i];
  const AimeScene S = sm.n ? sm.s[b] : scenes[b];
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

// -------------------------------------------------------------------------------------------------
// Pruning decisions of prune_merge (scenario_tree.py:293-327, 361-395) for every scene of a round, one wave per scene:
// path probability floor (cls * SCEN_PROB < 1e-3), ego end point against the target lane (distance - max-sigma > threshold),
// then the greedy merge in descending-probability order of modes whose topology signatures differ by <= pi/6 for every exo
// agent.  Same float32 expressions as the reference's per-mode loop; the signatures and end points are the ones k_aime_world
// just wrote.  sel[b][j] = mode index of the j-th kept mode (visiting order) or -1, sel_prob[b][j] its path probability.
// -------------------------------------------------------------------------------------------------
// (the decisions of scene b by one 64-lane wave: every lane ends with the same so / po)
__device__ __forceinline__ void aime_select_scene(const AimeScene &S, float sp, int b, int t, const float *__restrict__ cls,
                                                  const float *__restrict__ topo, const float *__restrict__ ego_end, int lane_check,
                                                  float dist_thres, float prob_floor, float (&so)[AIME_K], float (&po)[AIME_K]) {
  float cl[AIME_K], pr[AIME_K];
  bool keep[AIME_K];
#pragma unroll
  for (int k = 0; k < AIME_K; ++k) {
    cl[k] = cls[b * AIME_K + k];
    pr[k] = cl[k] * sp;
    keep[k] = !(pr[k] < prob_floor);
    if (lane_check) {
      const float *e = ego_end + ((size_t)b * AIME_K + k) * 4;
      if ((e[3] - e[2]) > dist_thres) keep[k] = false;
    }
  }
  // visiting rank of every mode: stable argsort of -cls
  int rank[AIME_K];
#pragma unroll
  for (int k = 0; k < AIME_K; ++k) {
    rank[k] = 0;
#pragma unroll
    for (int j = 0; j < AIME_K; ++j) rank[k] += (cl[j] > cl[k]) || (cl[j] == cl[k] && j < k);
  }
  // differ bit (k1, k2), k1 < k2: some exo agent's signatures differ by more than pi/6 (angle difference wrapped to (-pi, pi]:
  // d - 2 pi rint(d / 2 pi), within 1e-6 of the reference's atan2(sin d, cos d)) -- lanes stride over the agents
  const float thr = (float)(3.14159265358979323846 / 6.0);
  unsigned mask = 0;
  for (int i = S.a0 + 1 + t; i < S.a1; i += 64) {
    float sg[AIME_K];
#pragma unroll
    for (int k = 0; k < AIME_K; ++k) sg[k] = topo[(size_t)i * AIME_K + k];
#pragma unroll
    for (int k1 = 0; k1 < AIME_K; ++k1)
#pragma unroll
      for (int k2 = k1 + 1; k2 < AIME_K; ++k2) {
        float d = sg[k1] - sg[k2];
        d = d - 6.28318530717958647692f * rintf(d * 0.15915494309189533577f);
        if ((fabsf(d) - thr) > 0.f) mask |= 1u << (k1 * AIME_K + k2);
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mask |= __shfl_xor(mask, o, 64);
  // greedy merge over the kept modes in visiting order, all state in bit masks (no indexed arrays): the next selected mode is the
  // alive one of lowest rank; the modes that stay alive are those that differ from it
  unsigned alive = 0;
#pragma unroll
  for (int k = 0; k < AIME_K; ++k) alive |= keep[k] ? 1u << k : 0u;
#pragma unroll
  for (int k = 0; k < AIME_K; ++k) { so[k] = -1.f; po[k] = 0.f; }
  int ns = 0;
#pragma unroll
  for (int it = 0; it < AIME_K; ++it) {
    int best = -1, brank = AIME_K;
    float bp = 0.f;
    unsigned brow = 0;
#pragma unroll
    for (int k = 0; k < AIME_K; ++k)
      if (((alive >> k) & 1u) && rank[k] < brank) {
        best = k; brank = rank[k]; bp = pr[k];
        // symmetric row of the pair matrix for mode k: bit j set <=> (k, j) differ
        unsigned row = 0;
#pragma unroll
        for (int j = 0; j < AIME_K; ++j) {
          const int lo = k < j ? k : j, hi = k < j ? j : k;
          if (j != k && ((mask >> (lo * AIME_K + hi)) & 1u)) row |= 1u << j;
        }
        brow = row;
      }
    if (best >= 0) {
#pragma unroll
      for (int q = 0; q < AIME_K; ++q) if (q == ns) { so[q] = (float)best; po[q] = bp; }      // (no indexed register arrays)
      ++ns;
      alive &= brow;            // drops the selected mode itself (its own bit is 0) and everything merged into it
    }
  }
}

__global__ __launch_bounds__(64) void k_aime_select(const AimeScene *__restrict__ scenes, const float *__restrict__ cls,
                                                    const float *__restrict__ scen_prob, const float *__restrict__ topo,
                                                    const float *__restrict__ ego_end, int lane_check, float dist_thres,
                                                    float *__restrict__ sel, float *__restrict__ sel_prob, float prob_floor, const AimeSmall sm) {
  const int b = blockIdx.x, t = threadIdx.x;
  const AimeScene S = sm.n ? sm.s[b] : scenes[b];
  const float sp = sm.n ? sm.prob[b] : scen_prob[b];
  float so[AIME_K], po[AIME_K];
  aime_select_scene(S, sp, b, t, cls, topo, ego_end, lane_check, dist_thres, prob_floor, so, po);
  if (t != 0) return;
#pragma unroll
  for (int k = 0; k < AIME_K; ++k) { sel[(size_t)b * AIME_K + k] = so[k]; sel_prob[(size_t)b * AIME_K + k] = po[k]; }
}

// -------------------------------------------------------------------------------------------------
// Re-basing of the observation at a branch point (scenario_tree.py:467-567 update_obser, with
// utils.py:171-190 get_new_lane_graph / get_origin_rotation, :113-134 actor features, scenario_tree.py:
// 613-652 get_high_level_command, utils.py:193-242 get_rpe): from the last 50 world-frame steps of every
// agent of a child scene to the predictor inputs of the next AIME round, written where the predictor reads
// them.  One workgroup per child scene; float32 like the reference.
// -------------------------------------------------------------------------------------------------
#define RB_THREADS 256
#define RB_FEAT_BLOCKS(a) ((int)(((size_t)(a) * 48 + RB_THREADS - 1) / RB_THREADS))     // feature blocks per scene (k_aime_rebase grid y = 1 + this)
#define RB_T 50
#define RB_LANE_LDS 1024   // target-lane points staged in LDS by k_aime_rebase's frame block (longer lanes are walked in global memory)

struct RebaseArgs {
  int a, l, n_lane, pad_ones;
  const float *pos, *ang, *vel;       // [S,a,50,2], [S,a,50], [S,a,50,2]   world frame windows
  const float *types;                 // [a,50,7]
  const float *pad;                   // [S,a,50] or null (ones)
  const float *lane_ctrs0, *lane_vecs0;   // [l,2] lane anchors as the lane graph holds them
  const float *tlane, *tinfo;         // target lane [n_lane,2], per-point info [n_lane,12]
  float time_ahead, min_vel;
  float *actors, *actor_ctrs, *actor_vecs, *lane_ctrs, *lane_vecs, *tgt_nodes, *tgt_rpe, *frames;
  float travel0;            // >= 0: the look-ahead distance of get_high_level_command given by the caller (root scene: the ego speed comes
                            // from the ego state, scenario_tree.py:131, not from the window); < 0: max(|v_ego|, min_vel) * time_ahead
};

__device__ __forceinline__ void rb_cs(float ax, float ay, float bx, float by, float &c, float &s) {
  const float den = sqrtf(ax * ax + ay * ay) * sqrtf(bx * bx + by * by) + 1e-10f;
  c = (ax * bx + ay * by) / den;
  s = (ax * by - ay * bx) / den;
}

// Window of a child scene assembled on the device: the last 50 steps of [parent window (50) | the child's first `dur` predicted
// steps] -- what ChildScene.window6 / update_obser cut out on the host (scenario_tree.py:396-412, 470-473).  One block per
// (scene, agent), lane = window step.  prev_*: the previous re-basing call's window arrays ([S_prev, a, 50, .]).
// rows: [., 60, 6] world-frame rows; agent i of child s is row row0[s] + i * row_stride (1: rows gathered per kept mode, 6: k_aime_world's
// [A,6,60,6] buffer with row0 = first agent * 6 + mode).  cov_last (optional) receives the window's last max-sigma per agent
// (= TRAJS_COV_HIST[:, -1, 0] of the re-based scene: the child's own step dur - 1).
__global__ __launch_bounds__(64) void k_aime_windows(const float *__restrict__ prev_pos, const float *__restrict__ prev_ang,
                                                     const float *__restrict__ prev_vel, const float *__restrict__ rows,
                                                     const int *__restrict__ parent_slot, const int *__restrict__ row0,
                                                     const int *__restrict__ dur, int a, float *__restrict__ pos,
                                                     float *__restrict__ ang, float *__restrict__ vel, int row_stride,
                                                     float *__restrict__ cov_last) {
  const int s = blockIdx.x / a, i = blockIdx.x - s * a, t = threadIdx.x;
  if (t >= RB_T) return;
  const int d = dur[s];
  const int src = t + d;                         // index into [parent window | child steps]
  float x, y, vx, vy, h;
  if (src < RB_T) {
    const size_t p = ((size_t)parent_slot[s] * a + i) * RB_T + src;
    x = prev_pos[2 * p]; y = prev_pos[2 * p + 1]; vx = prev_vel[2 * p]; vy = prev_vel[2 * p + 1]; h = prev_ang[p];
  } else {
    const float *r = rows + (((size_t)row0[s] + (size_t)i * row_stride) * AIME_T + (src - RB_T)) * AIME_PK;
    x = r[0]; y = r[1]; vx = r[2]; vy = r[3]; h = r[4];
    if (cov_last && t == RB_T - 1) cov_last[(size_t)s * a + i] = r[5];
  }
  const size_t o = ((size_t)s * a + i) * RB_T + t;
  pos[2 * o] = x; pos[2 * o + 1] = y; vel[2 * o] = vx; vel[2 * o + 1] = vy; ang[o] = h;
}

// Branch-time test of decide_branch / get_branch_time (scenario_tree.py:82-100, 592-611) for every kept mode of a round: bit t of
// hit[b][j] (two 32-bit words) is set when some agent's predicted max-sigma at step t exceeds 9 x its max-sigma at the scene's
// compare step (float32 division and comparison, as the reference).  The host picks the first even t inside (CUR_T, END_T).
// One wave per (scene, kept slot), lane = step; sel as written by k_aime_select.
// h_sel != null (unsharded plans): this last kernel of a round also writes the round's decisions -- kept mode, its probability, the hit bits --
// where the HOST reads them (page-locked, mapped staging), so that no copy stands between the kernel's end and the host's replay.
__global__ __launch_bounds__(64) void k_aime_branch(const AimeScene *__restrict__ scenes, const float *__restrict__ sel,
                                                    const float *__restrict__ world, unsigned *__restrict__ hit,
                                                    const float *__restrict__ sel_prob, float *__restrict__ h_sel, float *__restrict__ h_selp,
                                                    unsigned *__restrict__ h_hit, const AimeSmall sm) {
  const int b = blockIdx.x / AIME_K, j = blockIdx.x % AIME_K, t = threadIdx.x;
  const AimeScene S = sm.n ? sm.s[b] : scenes[b];
  const int k = (int)sel[(size_t)b * AIME_K + j];
  bool h = false;
  if (k >= 0 && t < AIME_T) {
    for (int i = S.a0; i < S.a1; ++i) {
      const float *r = world + ((size_t)i * AIME_K + k) * AIME_T * AIME_PK;
      h |= (r[t * AIME_PK + 5] / r[S.cmp * AIME_PK + 5]) > 9.0f;
    }
  }
  const unsigned long long m = __ballot(h);
  if (t == 0) { hit[2 * (size_t)blockIdx.x] = (unsigned)m; hit[2 * (size_t)blockIdx.x + 1] = (unsigned)(m >> 32); }
  if (h_sel && t == 0) {
    h_hit[2 * (size_t)blockIdx.x] = (unsigned)m; h_hit[2 * (size_t)blockIdx.x + 1] = (unsigned)(m >> 32);
    h_sel[blockIdx.x] = sel[blockIdx.x]; h_selp[blockIdx.x] = sel_prob[blockIdx.x];
  }
}

// k_aime_select + k_aime_branch in ONE launch (mind_aime_plan): a block per (scene, kept slot) works out its scene's decisions itself -- the
// same code on the same inputs as k_aime_select, six times per scene, a few hundred instructions -- and then runs its slot's branch-time
// test with the agents' two values requested eight agents at a time (they were 16 dependent round trips).  Block (b, 0) writes the scene's
// decisions (device buffer + the host mirror).  Same expressions per item: the same bits as the two kernels.
__global__ __launch_bounds__(64) void k_aime_select_branch(const AimeScene *__restrict__ scenes, const float *__restrict__ cls,
                                                           const float *__restrict__ scen_prob, const float *__restrict__ topo,
                                                           const float *__restrict__ ego_end, int lane_check, float dist_thres, float prob_floor,
                                                           const float *__restrict__ world, float *__restrict__ sel, float *__restrict__ sel_prob,
                                                           unsigned *__restrict__ hit, float *__restrict__ h_sel, float *__restrict__ h_selp,
                                                           unsigned *__restrict__ h_hit, const AimeSmall sm) {
  const int b = blockIdx.x / AIME_K, j = blockIdx.x % AIME_K, t = threadIdx.x;
  const AimeScene S = sm.n ? sm.s[b] : scenes[b];
  const float sp = sm.n ? sm.prob[b] : scen_prob[b];
  float so[AIME_K], po[AIME_K];
  aime_select_scene(S, sp, b, t, cls, topo, ego_end, lane_check, dist_thres, prob_floor, so, po);
  float kf = -1.f, pj = 0.f;
#pragma unroll
  for (int q = 0; q < AIME_K; ++q) if (q == j) { kf = so[q]; pj = po[q]; }
  const int k = (int)kf;
  bool h = false;
  if (k >= 0 && t < AIME_T) {
    for (int i0 = S.a0; i0 < S.a1; i0 += 8) {
      float num[8], den[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = i0 + q < S.a1 ? i0 + q : S.a1 - 1;
        const float *r = world + ((size_t)i * AIME_K + k) * AIME_T * AIME_PK;
        num[q] = r[t * AIME_PK + 5]; den[q] = r[S.cmp * AIME_PK + 5];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) if (i0 + q < S.a1) h |= (num[q] / den[q]) > 9.0f;
    }
  }
  const unsigned long long m = __ballot(h);
  if (t == 0) {
    hit[2 * (size_t)blockIdx.x] = (unsigned)m; hit[2 * (size_t)blockIdx.x + 1] = (unsigned)(m >> 32);
    sel[blockIdx.x] = kf; sel_prob[blockIdx.x] = pj;
    if (h_sel) {
      h_hit[2 * (size_t)blockIdx.x] = (unsigned)m; h_hit[2 * (size_t)blockIdx.x + 1] = (unsigned)(m >> 32);
      h_sel[blockIdx.x] = kf; h_selp[blockIdx.x] = pj;
    }
  }
}

// Rows of the returned scenario trees (get_scenario_tree, scenario_tree.py:208-272): the first `dur` predicted steps (x, y, max-sigma)
// of every agent of the listed nodes, packed [a, dur, 3] per node.  job = {first row in world (agent 0), dur, destination offset
// in floats, agents}; one block per (job, agent).
struct AimeGather { int row0, dur, dst, a; };
__global__ __launch_bounds__(64) void k_aime_gather(const AimeGather *__restrict__ jobs, const int *__restrict__ job_of_block,
                                                    const int *__restrict__ agent_of_block, const float *const *__restrict__ world_of_job,
                                                    float *__restrict__ out) {
  const AimeGather J = jobs[job_of_block[blockIdx.x]];
  const int i = agent_of_block[blockIdx.x], t = threadIdx.x;
  if (t >= J.dur) return;
  const float *r = world_of_job[job_of_block[blockIdx.x]] + (((size_t)J.row0 + (size_t)i * AIME_K) * AIME_T + t) * AIME_PK;
  float *o = out + (size_t)J.dst + ((size_t)i * J.dur + t) * 3;
  o[0] = r[0]; o[1] = r[1]; o[2] = r[5];
}

// Flattened cost-tree rows (trajectory_tree.py:58-124 reads every even step of a scenario node's window): for the listed nodes the
// steps 0, 2, 4 ... < dur of every agent, mean [n, a, 2] and max-sigma [n, a] at the node's offset.  One block per (job, agent).
struct AimeFlat { int row0, n, dst, a; };
__global__ __launch_bounds__(64) void k_aime_flat(const AimeFlat *__restrict__ jobs, const int *__restrict__ job_of_block,
                                                  const int *__restrict__ agent_of_block, const float *const *__restrict__ world_of_job,
                                                  float *__restrict__ mean, float *__restrict__ cov) {
  const AimeFlat J = jobs[job_of_block[blockIdx.x]];
  const int i = agent_of_block[blockIdx.x], m = threadIdx.x;
  if (m >= J.n) return;
  const float *r = world_of_job[job_of_block[blockIdx.x]] + (((size_t)J.row0 + (size_t)i * AIME_K) * AIME_T + 2 * m) * AIME_PK;
  const size_t o = (size_t)(J.dst + m) * J.a + i;
  mean[2 * o] = r[0]; mean[2 * o + 1] = r[1];
  cov[o] = r[5];
}

// rows [n,128] repeated `times` times (LaneNet's output shared by every scene of a round)
__global__ void k_repeat_rows(const float *__restrict__ src, size_t n, int times, float *__restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * (size_t)times) return;
  dst[i] = src[i % n];
}

__global__ __launch_bounds__(RB_THREADS) void k_aime_rebase(RebaseArgs A) {
  const int sc = blockIdx.x, tid = threadIdx.x;
  const int a = A.a;
  extern __shared__ float rb_sm[];          // per agent: ctr x, ctr y, th, cos, sin (5 floats)
  __shared__ float fr[8];                   // orig x, y, theta, cos, sin, cur_vel
  __shared__ float rdv[RB_THREADS];
  __shared__ int rdi[RB_THREADS];
  __shared__ int s_idx;
  __shared__ float ctr[11][2];
  const float *pos = A.pos + (size_t)sc * a * RB_T * 2;
  const float *ang = A.ang + (size_t)sc * a * RB_T;
  const float *vel = A.vel + (size_t)sc * a * RB_T * 2;
  if (tid == 0) {
    const float th = ang[RB_T - 1];
    fr[0] = pos[(RB_T - 1) * 2]; fr[1] = pos[(RB_T - 1) * 2 + 1]; fr[2] = th; fr[3] = cosf(th); fr[4] = sinf(th);
  }
  __syncthreads();
  const float ox = fr[0], oy = fr[1], th0 = fr[2], c0 = fr[3], s0 = fr[4];
  // ---- per-agent frames (AV frame first, then the agent's own last pose)
  for (int i = tid; i < a; i += RB_THREADS) {
    const float dx = pos[(i * RB_T + RB_T - 1) * 2] - ox, dy = pos[(i * RB_T + RB_T - 1) * 2 + 1] - oy;
    const float cx = dx * c0 + dy * s0, cy = dx * (-s0) + dy * c0;
    const float thi = ang[i * RB_T + RB_T - 1] - th0;
    const float ci = cosf(thi), si = sinf(thi);
    float *o = rb_sm + 5 * i;
    o[0] = cx; o[1] = cy; o[2] = thi; o[3] = ci; o[4] = si;
    if (blockIdx.y == 0) {
      A.actor_ctrs[((size_t)sc * a + i) * 2] = cx; A.actor_ctrs[((size_t)sc * a + i) * 2 + 1] = cy;
      A.actor_vecs[((size_t)sc * a + i) * 2] = ci; A.actor_vecs[((size_t)sc * a + i) * 2 + 1] = si;
    }
  }
  __syncthreads();
  // grid (scenes, 1 + RB_FEAT_BLOCKS(a)): block y = 0 of a scene writes its frames, lane anchors and target window, blocks y >= 1 one slice
  // of the [a,14,48] actor features each (every block derives the per-agent frames it needs itself: same expressions, same bits) -- the
  // kernel sits on the plan's critical path between two AIME rounds (34 us as one workgroup per scene)
  const int fb = (int)blockIdx.y - 1, nfb = (int)gridDim.y - 1;
  // ---- actor features [a,14,48]: displacement, heading cos/sin, velocity, type one-hot, pad flag; steps 2..49
  for (int e = (nfb > 0 ? fb * RB_THREADS : 0) + tid; nfb > 0 ? (fb >= 0 && e < min(a * 48, (fb + 1) * RB_THREADS)) : e < a * 48; e += RB_THREADS) {
    const int i = e / 48, t = e % 48 + 2;
    const float *f = rb_sm + 5 * i;
    float pn[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int tt = t - 1 + q;
      const float dx = pos[(i * RB_T + tt) * 2] - ox, dy = pos[(i * RB_T + tt) * 2 + 1] - oy;
      const float x1 = (dx * c0 + dy * s0) - f[0], y1 = (dx * (-s0) + dy * c0) - f[1];
      pn[q][0] = x1 * f[3] + y1 * f[4];
      pn[q][1] = x1 * (-f[4]) + y1 * f[3];
    }
    const float an = (ang[i * RB_T + t] - th0) - f[2];
    const float vx0 = vel[(i * RB_T + t) * 2], vy0 = vel[(i * RB_T + t) * 2 + 1];
    const float vx1 = vx0 * c0 + vy0 * s0, vy1 = vx0 * (-s0) + vy0 * c0;
    float *o = A.actors + ((size_t)sc * a + i) * 14 * 48 + (t - 2);
    o[0 * 48] = pn[1][0] - pn[0][0];
    o[1 * 48] = pn[1][1] - pn[0][1];
    o[2 * 48] = cosf(an);
    o[3 * 48] = sinf(an);
    o[4 * 48] = vx1 * f[3] + vy1 * f[4];
    o[5 * 48] = vx1 * (-f[4]) + vy1 * f[3];
#pragma unroll
    for (int k = 0; k < 7; ++k) o[(6 + k) * 48] = A.types[(i * RB_T + t) * 7 + k];
    o[13 * 48] = A.pad ? A.pad[((size_t)sc * a + i) * RB_T + t] : 1.0f;
  }
  if (fb >= 0) return;
  // ---- lane anchors in the new frame (utils.py:171-177)
  for (int q = tid; q < A.l; q += RB_THREADS) {
    const float dx = A.lane_ctrs0[2 * q] - ox, dy = A.lane_ctrs0[2 * q + 1] - oy;
    A.lane_ctrs[((size_t)sc * A.l + q) * 2] = dx * c0 + dy * s0;
    A.lane_ctrs[((size_t)sc * A.l + q) * 2 + 1] = dx * (-s0) + dy * c0;
    const float vx = A.lane_vecs0[2 * q], vy = A.lane_vecs0[2 * q + 1];
    A.lane_vecs[((size_t)sc * A.l + q) * 2] = vx * c0 + vy * s0;
    A.lane_vecs[((size_t)sc * A.l + q) * 2 + 1] = vx * (-s0) + vy * c0;
  }
  // ---- high-level command: target-lane window ahead of the ego (scenario_tree.py:613-652)
  {
    // the target lane's points in LDS: thread 0 walks them one after the other below (a dependent global load per step -- some thirty of
    // them at 10 m/s -- made this block the longest of the kernel, and ActorNet waits for the kernel); same values, same operations
    __shared__ float tl_sm[2 * RB_LANE_LDS];
    const bool tl_lds = A.n_lane <= RB_LANE_LDS;
    if (tl_lds) {
      for (int q = tid; q < 2 * A.n_lane; q += RB_THREADS) tl_sm[q] = A.tlane[q];
      __syncthreads();
    }
    const float *tlane = tl_lds ? tl_sm : A.tlane;
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int q = tid; q < A.n_lane; q += RB_THREADS) {
      const float dx = tlane[2 * q] - ox, dy = tlane[2 * q + 1] - oy;
      const float d = sqrtf(dx * dx + dy * dy);
      if (d < best) { best = d; bi = q; }            // strided scan keeps the lowest index per thread
    }
    rdv[tid] = best; rdi[tid] = bi;
    __syncthreads();
    for (int o = RB_THREADS / 2; o > 0; o >>= 1) {
      if (tid < o) {
        const float v2 = rdv[tid + o];
        const int i2 = rdi[tid + o];
        if (v2 < rdv[tid] || (v2 == rdv[tid] && i2 < rdi[tid])) { rdv[tid] = v2; rdi[tid] = i2; }   // np.argmin: first minimum
      }
      __syncthreads();
    }
    if (tid == 0) {
      // ego speed in its own frame = |vel_n[0, 49]|
      const float *f = rb_sm;
      const float vx0 = vel[(RB_T - 1) * 2], vy0 = vel[(RB_T - 1) * 2 + 1];
      const float vx1 = vx0 * c0 + vy0 * s0, vy1 = vx0 * (-s0) + vy0 * c0;
      const float vnx = vx1 * f[3] + vy1 * f[4], vny = vx1 * (-f[4]) + vy1 * f[3];
      const float cur_vel = sqrtf(vnx * vnx + vny * vny);
      float travel = A.travel0 >= 0.f ? A.travel0 : fmaxf(cur_vel, A.min_vel) * A.time_ahead;
      int idx = rdi[0];
      const int n = A.n_lane;
      while (idx < n - 1 && travel > 0.f) {
        ++idx;
        const float sx = tlane[2 * idx] - tlane[2 * idx - 2], sy = tlane[2 * idx + 1] - tlane[2 * idx - 1];
        travel = travel - sqrtf(sx * sx + sy * sy);
      }
      if (idx == n - 1) --idx;
      idx = idx < 5 ? 5 : idx;
      idx = idx > n - 6 ? n - 6 : idx;
      s_idx = idx;
      fr[5] = cur_vel;
    }
    __syncthreads();
    const int idx = s_idx;
    float *frm = A.frames + (size_t)sc * 28;
    if (tid < 11) {
      const int q = idx - 5 + tid;
      const float px = tlane[2 * q], py = tlane[2 * q + 1];
      frm[6 + 2 * tid] = px; frm[6 + 2 * tid + 1] = py;                 // TGT_PTS (world frame)
      const float dx = px - ox, dy = py - oy;
      ctr[tid][0] = dx * c0 + dy * s0;
      ctr[tid][1] = dx * (-s0) + dy * c0;
    }
    __syncthreads();
    // the per-point info of the ten window pieces: plain copies, one per thread
    if (tid >= 64 && tid < 64 + 120) {
      const int e = tid - 64, q = e / 12, k = e - 12 * q;
      A.tgt_nodes[(size_t)sc * 160 + q * 16 + 4 + k] = A.tinfo[(size_t)(idx - 5 + q + 1) * 12 + k];
    }
    if (tid == 0) {
      frm[0] = c0; frm[1] = -s0; frm[2] = s0; frm[3] = c0; frm[4] = ox; frm[5] = oy;      // ROT (row-major), ORIG
      float mx = 0.f, my = 0.f;
      for (int q = 0; q < 11; ++q) { mx += ctr[q][0]; my += ctr[q][1]; }
      mx = mx / 11.0f; my = my / 11.0f;
      const float dvx = ctr[10][0] - ctr[0][0], dvy = ctr[10][1] - ctr[0][1];
      const float nrm = sqrtf(dvx * dvx + dvy * dvy);
      const float ax = dvx / nrm, ay = dvy / nrm;
      float ln[11][2];
      for (int q = 0; q < 11; ++q) {
        const float x1 = ctr[q][0] - mx, y1 = ctr[q][1] - my;
        ln[q][0] = x1 * ax + y1 * ay;           // (x, y) @ [[ax, -ay], [ay, ax]]
        ln[q][1] = x1 * (-ay) + y1 * ax;
      }
      float *tn = A.tgt_nodes + (size_t)sc * 160;
      for (int q = 0; q < 10; ++q) {
        tn[q * 16 + 0] = (ln[q][0] + ln[q + 1][0]) / 2.0f;
        tn[q * 16 + 1] = (ln[q][1] + ln[q + 1][1]) / 2.0f;
        tn[q * 16 + 2] = ln[q + 1][0] - ln[q][0];
        tn[q * 16 + 3] = ln[q + 1][1] - ln[q][1];
      }
      // TGT_RPE = get_rpe([anchor, ego], [anchor direction, ego direction]) flattened [5][i][j]
      const float cxs[2] = {mx, rb_sm[0]}, cys[2] = {my, rb_sm[1]};
      const float vxs[2] = {ax, rb_sm[3]}, vys[2] = {ay, rb_sm[4]};
      float *tr = A.tgt_rpe + (size_t)sc * 20;
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
          const float dx = cxs[j] - cxs[i], dy = cys[j] - cys[i];
          const float dist = sqrtf(dx * dx + dy * dy);
          float c1, s1, c2, s2;
          rb_cs(vxs[j], vys[j], vxs[i], vys[i], c1, s1);
          rb_cs(vxs[j], vys[j], dx, dy, c2, s2);
          tr[0 * 4 + i * 2 + j] = c1; tr[1 * 4 + i * 2 + j] = s1; tr[2 * 4 + i * 2 + j] = c2; tr[3 * 4 + i * 2 + j] = s2;
          tr[4 * 4 + i * 2 + j] = dist * 2.0f / 100.0f;
        }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Root scene on the device (process_data, scenario_tree.py:122-206, and prepare_root_data :414-465): the agent part is
// k_aime_rebase on the raw world-frame windows (same normalisation, utils.py:245-342 / scenario_tree.py:128-158); here the lane graph
// of update_lane_graph_from_argo (utils.py:345-483) from the map's resampled world-frame polylines -- float64 like the host code, cast
// to float32 at the end -- and the root's world-frame histories as prepare_root_data reconstructs them from the NORMALISED arrays
// (a float32 round trip the later re-basings build on).
// -------------------------------------------------------------------------------------------------
// one block per lane piece, 11 resampled points each: lane_ctrs / lane_vecs [l,2] (AV frame), LANES [l,10,16]
__global__ __launch_bounds__(64) void k_aime_root_lanes(const double *__restrict__ pts, const int *__restrict__ flags, const float *__restrict__ frames,
                                                        float *__restrict__ lane_ctrs, float *__restrict__ lane_vecs, float *__restrict__ lanes) {
  __shared__ double c[11][2], inst[11][2];
  __shared__ double anch[4];
  const int q = blockIdx.x, t = threadIdx.x;
  // frames: ROT row-major (c0, -s0, s0, c0), ORIG
  const double r00 = (double)frames[0], r01 = (double)frames[1], r10 = (double)frames[2], r11 = (double)frames[3];
  const double ox = (double)frames[4], oy = (double)frames[5];
  if (t < 11) {
    const double x = pts[((size_t)q * 11 + t) * 2] - ox, y = pts[((size_t)q * 11 + t) * 2 + 1] - oy;
    c[t][0] = x * r00 + y * r10;
    c[t][1] = x * r01 + y * r11;
  }
  __syncthreads();
  if (t == 0) {
    double mx = 0.0, my = 0.0;
    for (int k = 0; k < 11; ++k) { mx += c[k][0]; my += c[k][1]; }
    mx /= 11.0; my /= 11.0;
    const double dvx = c[10][0] - c[0][0], dvy = c[10][1] - c[0][1];
    const double nrm = sqrt(dvx * dvx + dvy * dvy);
    anch[0] = mx; anch[1] = my; anch[2] = dvx / nrm; anch[3] = dvy / nrm;
    lane_ctrs[2 * q] = (float)mx; lane_ctrs[2 * q + 1] = (float)my;
    lane_vecs[2 * q] = (float)anch[2]; lane_vecs[2 * q + 1] = (float)anch[3];
  }
  __syncthreads();
  if (t < 11) {
    const double x1 = c[t][0] - anch[0], y1 = c[t][1] - anch[1];
    inst[t][0] = x1 * anch[2] + y1 * anch[3];             // (x, y) @ [[ax, -ay], [ay, ax]]
    inst[t][1] = x1 * (-anch[3]) + y1 * anch[2];
  }
  __syncthreads();
  if (t < 10) {
    float *o = lanes + ((size_t)q * 10 + t) * 16;
    const int *f = flags + (size_t)q * 6;       // lane type (0..2), intersection, left mark class, right mark class, has left / right neighbour
    o[0] = (float)((inst[t][0] + inst[t + 1][0]) / 2.0);
    o[1] = (float)((inst[t][1] + inst[t + 1][1]) / 2.0);
    o[2] = (float)(inst[t + 1][0] - inst[t][0]);
    o[3] = (float)(inst[t + 1][1] - inst[t][1]);
    o[4] = (float)f[1];
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[5 + k] = f[0] == k ? 1.f : 0.f; o[8 + k] = f[2] == k ? 1.f : 0.f; o[11 + k] = f[3] == k ? 1.f : 0.f; }
    o[14] = (float)f[4]; o[15] = (float)f[5];
  }
}

// one block per agent, lane = history step: the root's world-frame windows [a,50,.] from the normalised arrays (float32, the
// expressions of k_aime_rebase for the normalisation and of prepare_root_data / _to_world for the way back), cov_last = 1e-5
__global__ __launch_bounds__(64) void k_aime_root_hist(const float *__restrict__ pos, const float *__restrict__ ang, const float *__restrict__ vel,
                                                       const float *__restrict__ frames, const float *__restrict__ actor_ctrs,
                                                       const float *__restrict__ actor_vecs, float *__restrict__ wpos, float *__restrict__ wang,
                                                       float *__restrict__ wvel, float *__restrict__ cov_last) {
  const int i = blockIdx.x, t = threadIdx.x;
  if (t >= RB_T) return;
  const float c0 = frames[0], s0 = frames[2], ox = frames[4], oy = frames[5];
  const float th0 = ang[RB_T - 1];                         // ego heading at the last step (agent 0)
  const float cx = actor_ctrs[2 * i], cy = actor_ctrs[2 * i + 1], ci = actor_vecs[2 * i], si = actor_vecs[2 * i + 1];
  const float thi = ang[i * RB_T + RB_T - 1] - th0;
  // normalised position / heading / velocity (normalize_agents)
  const float dx = pos[(i * RB_T + t) * 2] - ox, dy = pos[(i * RB_T + t) * 2 + 1] - oy;
  const float x1 = (dx * c0 + dy * s0) - cx, y1 = (dx * (-s0) + dy * c0) - cy;
  const float pnx = x1 * ci + y1 * si, pny = x1 * (-si) + y1 * ci;
  const float an = (ang[i * RB_T + t] - th0) - thi;
  const float vx0 = vel[(i * RB_T + t) * 2], vy0 = vel[(i * RB_T + t) * 2 + 1];
  const float vx1 = vx0 * c0 + vy0 * s0, vy1 = vx0 * (-s0) + vy0 * c0;
  const float vnx = vx1 * ci + vy1 * si, vny = vx1 * (-si) + vy1 * ci;
  // back to the world frame (prepare_root_data): R_i = rot(atan2(vec_i)), then the scene rotation / origin
  const float th = atan2f(si, ci);
  const float c = cosf(th), s = sinf(th);
  AimeScene S;
  S.r00 = c0; S.r01 = -s0; S.r10 = s0; S.r11 = c0; S.ox = ox; S.oy = oy;
  float wx, wy, wvx, wvy;
  aime_to_world(pnx, pny, c, s, cx, cy, S, true, wx, wy);
  aime_to_world(vnx, vny, c, s, 0.f, 0.f, S, false, wvx, wvy);
  const size_t o = (size_t)i * RB_T + t;
  wpos[2 * o] = wx; wpos[2 * o + 1] = wy; wvel[2 * o] = wvx; wvel[2 * o + 1] = wvy;
  wang[o] = (an + th) + atan2f(S.r10, S.r00);
  if (t == 0) cov_last[i] = 1e-5f;
}
