// mind_aime_plan: ScenarioTreeGenerator.branch_aime (planners/mind/scenario_tree.py:38-58) in one call.  Included by mind_hip.hip
// (uses mind_ctx, ensure / fail / HIPCHK, mind_predict_batch and the kernels of aime_kernels.hip).
//
// Per round the device runs  predictor -> k_aime_world -> k_aime_select -> k_aime_branch  and the host reads ONE small buffer (kept
// modes, path probabilities, branch-time bits: 72 B per scene); create_nodes / decide_branch (scenario_tree.py:73-100) are restated
// here over plain arrays; the branching nodes' observations are re-based where they are (k_aime_windows from k_aime_world's rows
// and the parents' windows, k_aime_rebase) and the next predictor call is queued before the host looks at anything else.  Every
// small table goes through page-locked staging, so no copy waits for the stream to drain.  At the end one gather kernel packs the
// rows get_scenario_tree (scenario_tree.py:208-272) attaches to the nodes of finished branches.
//
// Sharded (mind_set_exchange, include/mind_hip.h): the scenes of a round are block-distributed over the ranks; a rank runs the device
// part of the round on its block only, the decisions (96 B per scene) are all-gathered, the bookkeeping below is replayed identically on
// every rank, the rank that holds a branching node's parent scene re-bases it; of the next round's inputs only the small per-scene frames
// (28 floats: the replicated tree's node records) + LaneNet's output travel in one all-gather, a re-based scene's inputs + history windows go
// by an all-to-all from the rank that re-based it to the rank whose block of the next round holds it (MIND_XCHG_ALLTOALLV: on a full tree the
// same rank except at block boundaries; packed / unpacked by k_copy_segs; a round whose ranges coincide on every rank is skipped); the final
// rows / cost-tree entries are completed by one all-reduce over zero-filled buffers.  world == 1 runs the same code without the exchanges.
#include <chrono>
namespace {

struct PlScene {            // an observation pushed through the predictor: the root or a re-based branch node
  int node;                 // internal tree node it belongs to (0 = root)
  float prob;
  int cur_t, end_t;
  float rot[4], orig[2], tgt[22];
};

struct PlNode {
  int round, scene, mode, parent, depth;
  int owner, lscene;        // sharded: the rank whose pl_world[round] holds the node's predicted rows, its scene index there
  float prob;
  int cur_t, end_t;
  bool branch, end, term, rebased;
  unsigned long long hit;   // bit t: some agent's sigma at predicted step t > 9 x its sigma at the compare step
  float tgt[22];
};

int pl_pin(mind_ctx *c, int which, size_t bytes) {      // (declared ahead of ilqr_impl in mind_hip.hip)
  if (bytes <= c->pl_pin_cap[which]) return MIND_OK;
  if (c->pl_pin[which]) (void)hipHostFree(c->pl_pin[which]);
  c->pl_pin[which] = nullptr; c->pl_pin_cap[which] = 0;
  // (page-locked staging follows the device buffers' policy: small ones double from 1 MB -- hipHostFree + hipHostMalloc cost a millisecond)
  const size_t want = bytes < ((size_t)64 << 20) ? std::max<size_t>(2 * bytes, (size_t)1 << 20) : bytes + bytes / 2 + 4096;
  if (hipHostMalloc(&c->pl_pin[which], want, hipHostMallocDefault) != hipSuccess)
    return fail(c, MIND_ENOMEM, "hipHostMalloc(%zu) failed", want);
  c->pl_pin_cap[which] = want;
  return MIND_OK;
}

// contiguous block [lo, hi) of n items for rank r of w (the first ranks take the remainder: parallel.Shard.block)
inline void pl_block(int n, int r, int w, int &lo, int &hi) {
  const int base = n / w, rem = n % w;
  lo = r * base + (r < rem ? r : rem);
  hi = lo + base + (r < rem ? 1 : 0);
}
inline int pl_owner(int n, int w, int b) {
  for (int r = 0; r < w; ++r) { int lo, hi; pl_block(n, r, w, lo, hi); if (b >= lo && b < hi) return r; }
  return w - 1;
}

struct CopySeg { const float *src; float *dst; long long n; };
// segment copies of the exchange packing: blockIdx.y = segment, blockIdx.x = 2048-float chunk of it
__global__ __launch_bounds__(256) void k_copy_segs(const CopySeg *__restrict__ segs) {
  const CopySeg S = segs[blockIdx.y];
  const long long i0 = (long long)blockIdx.x * 2048 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const long long i = i0 + k * 256;
    if (i < S.n) S.dst[i] = S.src[i];
  }
}

int pl_exchange(mind_ctx *c, int op, void *send, void *recv, size_t bytes) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int rc = c->xfn(c->xuser, op, send, recv, (int64_t)bytes);
  if (rc) return fail(c, MIND_EHIP, "mind_aime_plan: the exchange callback failed (%d)", rc);
  c->x_collectives += 1;
  c->x_bytes += op == MIND_XCHG_ALLGATHER ? (long long)bytes * c->xw : (long long)bytes;
  return MIND_OK;
}

// all-to-all with per-pair sizes: table = [2][world] bytes this rank sends to / receives from every rank (send / recv are packed in rank order)
int pl_exchange_v(mind_ctx *c, void *send, void *recv, const int64_t *table) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int rc = c->xfn(c->xuser, MIND_XCHG_ALLTOALLV, send, recv, (int64_t)(intptr_t)table);
  if (rc) return fail(c, MIND_EHIP, "mind_aime_plan: the exchange callback failed (%d)", rc);
  c->x_collectives += 1;
  for (int r = 0; r < c->xw; ++r) c->x_bytes += (long long)table[(size_t)c->xw + r];
  return MIND_OK;
}

int pl_copy_segs(mind_ctx *c, const std::vector<CopySeg> &segs) {
  if (segs.empty()) return MIND_OK;
  long long mx = 0;
  for (const CopySeg &s : segs) mx = std::max(mx, s.n);
  if (mx == 0) return MIND_OK;
  int rc;
  if ((rc = ensure(c, c->x_seg, segs.size() * sizeof(CopySeg)))) return rc;
  if ((rc = pl_pin(c, 6, segs.size() * sizeof(CopySeg)))) return rc;      // (the previous table's copy completed: an exchange = a stream synchronisation lies between two uses)
  memcpy(c->pl_pin[6], segs.data(), segs.size() * sizeof(CopySeg));
  HIPCHK(c, hipMemcpyAsync(c->x_seg.p, c->pl_pin[6], segs.size() * sizeof(CopySeg), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_copy_segs, dim3((unsigned)((mx + 2047) / 2048), (unsigned)segs.size()), dim3(256), 0, c->stream, (const CopySeg *)c->x_seg.p);
  HIPCHK(c, hipGetLastError());
  return MIND_OK;
}

}  // namespace

extern "C" int mind_aime_plan(mind_ctx *c, const mind_aime_plan_in *in, mind_aime_plan_out *out) {
  if (!c || !in || !out) return MIND_EINVAL;
  if (!c->have_weights) return fail(c, MIND_ESTATE, "weights not loaded");
  if (c->il_finish) {
    // A tree-iLQR call is pending on this context.  The solves a plan began itself write into the library's own pl_sol_* vectors and
    // staging tables, which this plan is about to resize: they are drained and dropped here (their caller gave them up -- an exception
    // between plan_start and plan_end, a planner that was reset).  A call begun by the caller (mind_ilqr_contingency_begin) writes into
    // the caller's arrays, whose lifetime the library cannot know: that one is refused.
    if (!c->il_finish_owned)
      return fail(c, MIND_ESTATE, "mind_aime_plan: a tree-iLQR call begun with mind_ilqr_contingency_begin is pending on this context (mind_ilqr_finish first)");
    (void)mind_ilqr_finish(c);
    c->pl_sol_xs.clear(); c->pl_sol_us.clear(); c->pl_sol_stw.clear(); c->pl_sol_stf.clear();
  }
  const int a = in->n_agents, l = in->n_lanes, P = in->n_lane_pts;
  const bool raw = in->raw_pos != nullptr;      // root scene featurised on the device
  if (a <= 0 || l <= 0 || P < 12 || !in->types || !in->target_lane || !in->target_lane_info ||
      (raw ? (!in->raw_ang || !in->raw_vel || !in->raw_pad || !in->lane_pts || !in->lane_flags || !(in->travel0 >= 0.f))
           : (!in->actors || !in->actor_ctrs || !in->actor_vecs || !in->lanes || !in->lane_ctrs || !in->lane_vecs || !in->tgt_nodes ||
              !in->tgt_rpe || !in->rot || !in->orig || !in->tgt_pts || !in->hist)) || in->max_depth < 0 || in->max_rounds <= 0 || in->max_rounds > 32 || in->pred_len < 2 || in->pred_len > AIME_T)
    return fail(c, MIND_EINVAL, "mind_aime_plan: bad argument");
  if (in->script_cls && (!in->script_reg || !in->script_vel)) return fail(c, MIND_EINVAL, "mind_aime_plan: scripted modes need cls, reg and vel");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  c->pl_plan_agents = 0; c->pl_tree_top.clear();      // (the previous plan's cost trees are gone whatever happens below)
  c->pl_gen += 1;
  if (!c->ev_pl) HIPCHK(c, hipEventCreateWithFlags(&c->ev_pl, hipEventDisableTiming));
  if (!c->ev_tab) HIPCHK(c, hipEventCreateWithFlags(&c->ev_tab, hipEventDisableTiming));
  if (!c->pl_copy) HIPCHK(c, hipStreamCreateWithFlags(&c->pl_copy, hipStreamNonBlocking));
  int rc;
  const int T = AIME_T, OBS = RB_T, HZ = in->pred_len;     // predicted steps per mode, history window, planning horizon
  // ---- root upload (one page-locked staging buffer -> one async copy), floats
  size_t o = 0;
  auto take = [&](size_t n) { const size_t r = o; o += (n + 3) & ~(size_t)3; return r; };
  const size_t o_actors = take((size_t)a * 14 * 48), o_ctrs = take((size_t)a * 2), o_vecs = take((size_t)a * 2), o_lanes = take((size_t)l * 160);
  const size_t o_lc = take((size_t)l * 2), o_lv = take((size_t)l * 2), o_tn = take(160), o_tr = take(20), o_cov = take(a);
  const size_t o_types = take((size_t)a * OBS * 7), o_tl = take((size_t)P * 2), o_ti = take((size_t)P * 12);
  const size_t o_wpos = take((size_t)a * OBS * 2), o_wang = take((size_t)a * OBS), o_wvel = take((size_t)a * OBS * 2);
  // (device-built root: raw windows / pad flags / lane polylines go up, the slots above are filled by kernels)
  const size_t o_fr = take(28), o_rpos = take(raw ? (size_t)a * OBS * 2 : 0), o_rang = take(raw ? (size_t)a * OBS : 0);
  const size_t o_rvel = take(raw ? (size_t)a * OBS * 2 : 0), o_rpad = take(raw ? (size_t)a * OBS : 0);
  const size_t o_lpts = take(raw ? (size_t)l * 11 * 2 * 2 : 0), o_lfl = take(raw ? (size_t)l * 6 : 0);      // doubles (2 floats each), ints
  const size_t n_root = o;
  if ((rc = ensure(c, c->pl_root, n_root * sizeof(float)))) return rc;
  if ((rc = pl_pin(c, 0, n_root * sizeof(float)))) return rc;
  if (raw) {
    float *h = (float *)c->pl_pin[0];
    memcpy(h + o_types, in->types, (size_t)a * OBS * 7 * sizeof(float));
    memcpy(h + o_tl, in->target_lane, (size_t)P * 2 * sizeof(float));
    memcpy(h + o_ti, in->target_lane_info, (size_t)P * 12 * sizeof(float));
    memcpy(h + o_rpos, in->raw_pos, (size_t)a * OBS * 2 * sizeof(float));
    memcpy(h + o_rang, in->raw_ang, (size_t)a * OBS * sizeof(float));
    memcpy(h + o_rvel, in->raw_vel, (size_t)a * OBS * 2 * sizeof(float));
    memcpy(h + o_rpad, in->raw_pad, (size_t)a * OBS * sizeof(float));
    memcpy(h + o_lpts, in->lane_pts, (size_t)l * 22 * sizeof(double));
    memcpy(h + o_lfl, in->lane_flags, (size_t)l * 6 * sizeof(int));
    // one copy: [types .. lane flags] (the slots before o_types are produced on the device)
    if ((rc = pl_upload(c, (float *)c->pl_root.p + o_types, h + o_types, (n_root - o_types) * sizeof(float), st))) return rc;
    float *d = (float *)c->pl_root.p;
    RebaseArgs R;
    R.a = a; R.l = 0; R.n_lane = P; R.pad_ones = 0;
    R.pos = d + o_rpos; R.ang = d + o_rang; R.vel = d + o_rvel; R.types = d + o_types; R.pad = d + o_rpad;
    R.lane_ctrs0 = nullptr; R.lane_vecs0 = nullptr; R.tlane = d + o_tl; R.tinfo = d + o_ti;
    R.time_ahead = in->time_ahead; R.min_vel = in->min_vel; R.travel0 = in->travel0;
    R.actors = d + o_actors; R.actor_ctrs = d + o_ctrs; R.actor_vecs = d + o_vecs; R.lane_ctrs = nullptr; R.lane_vecs = nullptr;
    R.tgt_nodes = d + o_tn; R.tgt_rpe = d + o_tr; R.frames = d + o_fr;
    hipLaunchKernelGGL(k_aime_rebase, dim3(1, 1 + RB_FEAT_BLOCKS(a)), dim3(RB_THREADS), 5 * (size_t)a * sizeof(float), st, R);
    // The lane graph, the root's world-frame histories and the frame read-back feed LaneNet / the token positions (side stream) and the
    // glue behind the predictor -- not ActorNet, which only needs k_aime_rebase's features: they go to the side stream, in front of the
    // predictor's own side-stream work, and ActorNet starts right behind the re-basing (they stood 25 us in front of it)
    hipStream_t rs = st;
    if (c->side && !(c->xfn && (c->xw > 1 || c->xforce))) {
      if (!c->ev_root) HIPCHK(c, hipEventCreateWithFlags(&c->ev_root, hipEventDisableTiming));
      HIPCHK(c, hipEventRecord(c->ev_root, st));
      HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_root, 0));
      rs = c->side;
    }
    hipLaunchKernelGGL(k_aime_root_lanes, dim3(l), dim3(64), 0, rs, (const double *)(d + o_lpts), (const int *)(d + o_lfl), (const float *)(d + o_fr),
                       d + o_lc, d + o_lv, d + o_lanes);
    hipLaunchKernelGGL(k_aime_root_hist, dim3(a), dim3(64), 0, rs, (const float *)(d + o_rpos), (const float *)(d + o_rang), (const float *)(d + o_rvel),
                       (const float *)(d + o_fr), (const float *)(d + o_ctrs), (const float *)(d + o_vecs), d + o_wpos, d + o_wang, d + o_wvel, d + o_cov);
    HIPCHK(c, hipGetLastError());
    if ((rc = pl_pin(c, 3, 28 * sizeof(float)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pl_pin[3], d + o_fr, 28 * sizeof(float), hipMemcpyDeviceToHost, rs));
    HIPCHK(c, hipEventRecord(c->ev_pl, rs));
  } else {
    float *h = (float *)c->pl_pin[0];
    memcpy(h + o_actors, in->actors, (size_t)a * 14 * 48 * sizeof(float));
    memcpy(h + o_ctrs, in->actor_ctrs, (size_t)a * 2 * sizeof(float));
    memcpy(h + o_vecs, in->actor_vecs, (size_t)a * 2 * sizeof(float));
    memcpy(h + o_lanes, in->lanes, (size_t)l * 160 * sizeof(float));
    memcpy(h + o_lc, in->lane_ctrs, (size_t)l * 2 * sizeof(float));
    memcpy(h + o_lv, in->lane_vecs, (size_t)l * 2 * sizeof(float));
    memcpy(h + o_tn, in->tgt_nodes, 160 * sizeof(float));
    memcpy(h + o_tr, in->tgt_rpe, 20 * sizeof(float));
    memcpy(h + o_types, in->types, (size_t)a * OBS * 7 * sizeof(float));
    memcpy(h + o_tl, in->target_lane, (size_t)P * 2 * sizeof(float));
    memcpy(h + o_ti, in->target_lane_info, (size_t)P * 12 * sizeof(float));
    for (int i = 0; i < a; ++i) {
      h[o_cov + i] = in->hist[((size_t)i * OBS + OBS - 1) * 6 + 5];           // TRAJS_COV_HIST[:, -1, 0]
      for (int t = 0; t < OBS; ++t) {
        const float *r = in->hist + ((size_t)i * OBS + t) * 6;
        h[o_wpos + ((size_t)i * OBS + t) * 2] = r[0]; h[o_wpos + ((size_t)i * OBS + t) * 2 + 1] = r[1];
        h[o_wvel + ((size_t)i * OBS + t) * 2] = r[2]; h[o_wvel + ((size_t)i * OBS + t) * 2 + 1] = r[3];
        h[o_wang + (size_t)i * OBS + t] = r[4];
      }
    }
    HIPCHK(c, hipMemcpyAsync(c->pl_root.p, h, n_root * sizeof(float), hipMemcpyHostToDevice, st));
  }
  const float *droot = (const float *)c->pl_root.p;
  if ((rc = ensure(c, c->pl_lf, (size_t)l * 128 * sizeof(float)))) return rc;
  // the lane-distance field of the contingency solves this plan will begin: everything it depends on is known now (ilqr_impl adopts it)
  if (in->solve_cfg_full && in->solve_x0 && in->solve_lane && in->solve_n_lane_pts >= 2 && !(c->xfn && (c->xw > 1 || c->xforce)) &&
      (rc = il_field_prepare(c, in->solve_cfg_full, in->solve_x0, in->solve_lane, in->solve_n_lane_pts, c->pl_copy)))
    return rc;

  // ---- internal tree (scenario_tree.py:60-67): root = node 0, a leaf with branch_flag
  std::vector<PlNode> nodes(1);
  {
    PlNode &r = nodes[0];
    memset(&r, 0, sizeof(r));
    r.round = -1; r.parent = -1; r.prob = 1.f; r.end_t = HZ; r.branch = true;
  }
  std::vector<int> leaves(1, 0);
  std::vector<PlScene> batch(1);
  {
    PlScene &s = batch[0];
    s.node = 0; s.prob = 1.f; s.cur_t = 0; s.end_t = HZ;
    if (!raw) { memcpy(s.rot, in->rot, 4 * sizeof(float)); memcpy(s.orig, in->orig, 2 * sizeof(float)); memcpy(s.tgt, in->tgt_pts, 22 * sizeof(float)); }
  }
  const float *prev_pos = droot + o_wpos, *prev_ang = droot + o_wang, *prev_vel = droot + o_wvel;
  const float *cov_last_dev = droot + o_cov;
  int cur_in = -1;                 // which pl_in holds the current round's predictor inputs (-1: the root upload)
  bool frames_pending = raw;       // device-built root: its frame (ROT, ORIG, TGT_PTS) comes back while the first predictor call runs
  int n_expanded = 0, round = 0;
  float pair_ms = 0.f;
  int pair_launches = 0;
  memset(out, 0, sizeof(*out));
  // MIND_PLAN_TRACE=1: host time stamps of this call's sections on stderr (diagnostic: where the host stands between the kernels)
  static const bool plan_trace = getenv("MIND_PLAN_TRACE") != nullptr;
  const auto tr_t0 = std::chrono::steady_clock::now();
  auto TR = [&](const char *what) {
    if (plan_trace) fprintf(stderr, "[plan] %8.1f us  %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr_t0).count(), what);
  };
  // profiling (bench.py's live pair-kernel durations): the predictor calls of the plan record into an event pool that is read once at
  // the end, instead of draining the stream after every call
  struct DeferGuard { mind_ctx *c; ~DeferGuard() { c->ev_defer = false; c->ev_pending.clear(); c->ev_pool_used = 0; } } defer_guard{c};
  c->ev_defer = c->profiling;
  c->ev_pending.clear(); c->ev_pool_used = 0;

  // layout of one re-based input set (floats): actors | actor_ctrs | actor_vecs | lane_ctrs | lane_vecs | tgt_nodes | tgt_rpe | frames | cov_last
  struct InOff { size_t actors, ctrs, vecs, lc, lv, tn, tr, fr, cov, total; };
  auto in_off = [&](size_t S) {
    InOff q; size_t p = 0;
    auto tk = [&](size_t n) { const size_t r = p; p += (n + 3) & ~(size_t)3; return r; };
    q.actors = tk(S * a * 14 * 48); q.ctrs = tk(S * a * 2); q.vecs = tk(S * a * 2); q.lc = tk(S * l * 2); q.lv = tk(S * l * 2);
    q.tn = tk(S * 160); q.tr = tk(S * 20); q.fr = tk(S * 28); q.cov = tk(S * a); q.total = p;
    return q;
  };

  const int XW = c->xfn ? c->xw : 1, XR = c->xfn ? c->xr : 0;
  const bool dist = c->xfn && (XW > 1 || c->xforce);       // exchanges run (a forced one-rank group included)
  std::vector<int> my_lo_of_round;                          // first scene of this rank's block, per round

  for (;; ++round) {
    const int B = (int)batch.size();
    if (round >= in->max_rounds) return fail(c, MIND_ESTATE, "unsupported: more than %d AIME rounds", in->max_rounds);
    int lo, hi;
    pl_block(B, XR, XW, lo, hi);
    const int Bk = hi - lo, A = Bk * a;                     // this rank's scenes [lo, hi) of the round, their agent rows
    const int Bmax = (B + XW - 1) / XW;                     // the largest block (the decision buffers are laid out for it)
    my_lo_of_round.push_back(lo);
    out->round_scenes[round] = Bk;
    // ---- this rank's scenes of the round through predictor -> k_aime_world -> k_aime_select -> k_aime_branch, in chunks of scenes when the
    //      round's edge tensor would not fit the budget (mind_set_tuning "plan_chunk_mb"): the scenes are independent, a chunk is a smaller launch
    // small outputs: topo [A,6] | ego_end [Bk,6,4] | decisions laid out for Bmax scenes: sel [Bmax,6] | sel_prob [Bmax,6] | hit [Bmax,6,2]
    const size_t n_topo = ((size_t)A * 6 + 3) & ~(size_t)3, n_ego = (size_t)Bk * 24, n_back = (size_t)Bmax * 6 * 4;
    if ((rc = ensure(c, c->pl_small, (n_topo + n_ego + n_back) * sizeof(float)))) return rc;
    float *d_topo = (float *)c->pl_small.p, *d_ego = d_topo + n_topo, *d_sel = d_ego + n_ego, *d_selp = d_sel + (size_t)Bmax * 6;
    unsigned *d_hit = (unsigned *)(d_selp + (size_t)Bmax * 6);
    // unsharded: the round's last kernel writes the decisions where the host reads them (same layout as the device buffer)
    float *h_mirror = nullptr;
    if (!dist && c->dec_mirror) {
      if ((rc = pl_pin(c, 2, n_back * sizeof(float)))) return rc;
      h_mirror = (float *)c->pl_pin[2];
    }
    DevBuf &tabb = c->pl_tab[round & 1];
    const size_t bS = ((size_t)Bk * sizeof(AimeScene) + 15) & ~(size_t)15, bI = ((size_t)A * sizeof(int) + 15) & ~(size_t)15;
    const size_t bP = ((size_t)Bk * sizeof(float) + 15) & ~(size_t)15;
    const size_t tab_bytes = bS + bI + bP;
    // (two table buffers, by round parity: the stream may still hold the previous round's windows kernel, which reads the index lists behind
    // that round's tables)
    if ((rc = ensure(c, tabb, tab_bytes + 3 * (size_t)6 * Bmax * sizeof(int) + 64))) return rc;
    if ((rc = pl_pin(c, 1, tab_bytes + 3 * (size_t)6 * Bmax * sizeof(int) + 64))) return rc;
    if ((int)c->pl_world.size() <= round) c->pl_world.resize(round + 1);
    float *d_world = nullptr;
    const int Ntok = a + l + 1;
    const double edge_mb = (double)Ntok * (((Ntok + 15) / 16) * 16) * (c->pair_prec == 2 && c->pair_tile ? 256.0 : 512.0) / (1024.0 * 1024.0);
    const int chunk = std::max(1, (int)std::min<double>((double)(Bk > 0 ? Bk : 1), (double)c->plan_chunk_mb / edge_mb));
    if (Bk > 0) {
      if ((rc = ensure(c, c->pl_world[round], (size_t)A * 6 * T * 6 * sizeof(float)))) return rc;
      d_world = (float *)c->pl_world[round].p;
      const int cbm = std::min(chunk, Bk);
      const size_t n_cls = ((size_t)cbm * 6 + 3) & ~(size_t)3;
      if ((rc = ensure(c, c->pl_pred, (n_cls + (size_t)cbm * a * 6 * T * 7) * sizeof(float)))) return rc;
    }
    // ---- the frames of the re-based scenes + the scene tables of the block: prepared BEHIND the launch of the round's first predictor
    //      call (which needs neither): waiting for the frames first put a host round trip between k_aime_rebase and the predictor
    bool tables_done = false;
    const bool all_small = c->tab_small && std::min(chunk, std::max(Bk, 1)) <= AIME_SMALL;
    auto prepare_tables = [&]() -> int {
      tables_done = true;
      // ---- the frames of the re-based scenes (ROT, ORIG, TGT_PTS; queued behind k_aime_rebase / the unpacking): needed by the scene tables
      if (frames_pending) {
        HIPCHK(c, hipEventSynchronize(c->ev_pl));
        const float *fr = (const float *)c->pl_pin[3];
        for (int b = 0; b < B; ++b) {
          memcpy(batch[b].rot, fr + (size_t)b * 28, 4 * sizeof(float));
          memcpy(batch[b].orig, fr + (size_t)b * 28 + 4, 2 * sizeof(float));
          memcpy(batch[b].tgt, fr + (size_t)b * 28 + 6, 22 * sizeof(float));
        }
        frames_pending = false;
      }
      if (Bk > 0) {
        // the scene tables of the whole block, one upload; agent rows are counted from the start of a scene's chunk
        char *h = (char *)c->pl_pin[1];
        AimeScene *hs = (AimeScene *)h;
        int *as = (int *)(h + bS);
        float *sp = (float *)(h + bS + bI);
        for (int b = 0; b < Bk; ++b) {
          AimeScene &S = hs[b];
          const PlScene &q = batch[lo + b];
          const int bc = b % chunk;
          S.a0 = a * bc; S.a1 = a * (bc + 1); S.last = HZ - 1;     /* seq_len - 1 - history length */ S.cmp = q.cur_t == 0 ? 1 : q.cur_t; S.pad2 = 0.f;
          S.r00 = q.rot[0]; S.r01 = q.rot[1]; S.r10 = q.rot[2]; S.r11 = q.rot[3];
          S.ox = q.orig[0]; S.oy = q.orig[1];
          S.theta_g = atan2f(S.r10, S.r00);
          for (int i = 0; i < a; ++i) as[(size_t)b * a + i] = bc;
          sp[b] = q.prob;
        }
        if (all_small) {
          // (the glue kernels take these tables by value: AimeSmall)
        } else if (c->pl_tab_side) {
          HIPCHK(c, hipMemcpyAsync(tabb.p, h, tab_bytes, hipMemcpyHostToDevice, c->pl_copy));
          HIPCHK(c, hipEventRecord(c->ev_tab, c->pl_copy));
          HIPCHK(c, hipStreamWaitEvent(st, c->ev_tab, 0));
        } else {
          HIPCHK(c, hipMemcpyAsync(tabb.p, h, tab_bytes, hipMemcpyHostToDevice, st));
        }
      }
      return MIND_OK;
    };
    const char *dtab = (const char *)tabb.p;
    for (int c0 = 0; c0 < Bk; c0 += chunk) {
      const int cb = std::min(chunk, Bk - c0), g0 = lo + c0, Ac = cb * a;      // scenes [g0, g0 + cb) of the round
      std::vector<int32_t> ao(cb + 1), lof(cb + 1);
      for (int b = 0; b <= cb; ++b) { ao[b] = a * b; lof[b] = l * b; }
      const size_t n_cls = ((size_t)cb * 6 + 3) & ~(size_t)3;
      float *d_cls = (float *)c->pl_pred.p, *d_reg = d_cls + n_cls, *d_vel = d_reg + (size_t)Ac * 6 * T * 5;
      const float *d_ctrs, *d_vecs;
      mind_scene_batch sb;
      memset(&sb, 0, sizeof(sb));
      mind_pred_out po;
      memset(&po, 0, sizeof(po));
      sb.n_scenes = cb; sb.actor_off = ao.data(); sb.lane_off = lof.data();
      if (cur_in < 0) {
        sb.actors = droot + o_actors; sb.lanes = droot + o_lanes; sb.actor_ctrs = d_ctrs = droot + o_ctrs; sb.actor_vecs = d_vecs = droot + o_vecs;
        sb.lane_ctrs = droot + o_lc; sb.lane_vecs = droot + o_lv; sb.tgt_nodes = droot + o_tn; sb.tgt_rpe = droot + o_tr;
        po.lane_feat = (float *)c->pl_lf.p;
      } else {
        const InOff q = in_off(B);
        const float *d = (const float *)c->pl_in[cur_in].p;
        sb.actors = d + q.actors + (size_t)g0 * a * 14 * 48; sb.actor_ctrs = d_ctrs = d + q.ctrs + (size_t)g0 * a * 2; sb.actor_vecs = d_vecs = d + q.vecs + (size_t)g0 * a * 2;
        sb.lane_ctrs = d + q.lc + (size_t)g0 * l * 2; sb.lane_vecs = d + q.lv + (size_t)g0 * l * 2;
        sb.tgt_nodes = d + q.tn + (size_t)g0 * 160; sb.tgt_rpe = d + q.tr + (size_t)g0 * 20;
        sb.lane_feat = cb == 1 ? (const float *)c->pl_lf.p : (const float *)c->pl_lrep.p;
      }
      po.cls = d_cls; po.reg = d_reg; po.vel = d_vel;
      TR("round: predictor launch begins");
      if ((rc = mind_predict_batch(c, &sb, &po))) return rc;
      TR("round: predictor launched");
      if (!tables_done && (rc = prepare_tables())) return rc;
      TR("round: tables prepared");
      n_expanded += cb;
      if (c->profiling) pair_launches += c->n_pair_launch;
      if (in->script_cls) {
        // scripted modes (benchmark hook): the forward above was the timed work, its outputs are replaced scene by scene
        const size_t nr = (size_t)a * 6 * T * 5, nv = (size_t)a * 6 * T * 2;
        hipLaunchKernelGGL(k_repeat_rows, dim3((unsigned)((6 * (size_t)cb + 255) / 256)), dim3(256), 0, st, in->script_cls, (size_t)6, cb, d_cls);
        hipLaunchKernelGGL(k_repeat_rows, dim3((unsigned)((nr * cb + 255) / 256)), dim3(256), 0, st, in->script_reg, nr, cb, d_reg);
        hipLaunchKernelGGL(k_repeat_rows, dim3((unsigned)((nv * cb + 255) / 256)), dim3(256), 0, st, in->script_vel, nv, cb, d_vel);
      }
      // prune_merge arithmetic + decisions + branch-time bits of the chunk, written at the chunk's place in the block's buffers
      const AimeScene *t_sc = (const AimeScene *)dtab + c0;
      float *w_c = d_world + (size_t)c0 * a * 6 * T * 6;
      AimeSmall sm;
      memset(&sm, 0, sizeof(sm));
      if (all_small) {
        const char *h = (const char *)c->pl_pin[1];
        memcpy(sm.s, (const AimeScene *)h + c0, (size_t)cb * sizeof(AimeScene));
        memcpy(sm.prob, (const float *)(h + bS + bI) + c0, (size_t)cb * sizeof(float));
        sm.n = cb; sm.a = a;
      }
      hipLaunchKernelGGL(k_aime_world, dim3(Ac * AIME_K), dim3(64), 0, st, t_sc, (const int *)(dtab + bS) + (size_t)c0 * a, d_reg, d_vel, d_ctrs, d_vecs,
                         cov_last_dev + (size_t)g0 * a, w_c, d_topo + (size_t)c0 * a * 6, d_ego + (size_t)c0 * 24, droot + o_tl, P, sm);
      float *hs_ = h_mirror ? h_mirror + (size_t)c0 * 6 : nullptr, *hp_ = h_mirror ? h_mirror + (size_t)Bmax * 6 + (size_t)c0 * 6 : nullptr;
      unsigned *hh_ = h_mirror ? (unsigned *)(h_mirror + (size_t)Bmax * 12) + (size_t)c0 * 12 : nullptr;
      const float pf_ = in->prob_floor > 0.f ? in->prob_floor : 0.001f;
      if (c->glue_fused) {
        // pruning decisions + branch-time bits in one launch (every block derives its scene's decisions itself)
        hipLaunchKernelGGL(k_aime_select_branch, dim3(cb * AIME_K), dim3(64), 0, st, t_sc, d_cls, (const float *)(dtab + bS + bI) + c0, d_topo + (size_t)c0 * a * 6,
                           d_ego + (size_t)c0 * 24, 1, in->dist_thres, pf_, w_c, d_sel + (size_t)c0 * 6, d_selp + (size_t)c0 * 6, d_hit + (size_t)c0 * 12, hs_, hp_, hh_, sm);
      } else {
        hipLaunchKernelGGL(k_aime_select, dim3(cb), dim3(64), 0, st, t_sc, d_cls, (const float *)(dtab + bS + bI) + c0, d_topo + (size_t)c0 * a * 6,
                           d_ego + (size_t)c0 * 24, 1, in->dist_thres, d_sel + (size_t)c0 * 6, d_selp + (size_t)c0 * 6, pf_, sm);
        hipLaunchKernelGGL(k_aime_branch, dim3(cb * AIME_K), dim3(64), 0, st, t_sc, d_sel + (size_t)c0 * 6, w_c, d_hit + (size_t)c0 * 12,
                           (const float *)d_selp + (size_t)c0 * 6, hs_, hp_, hh_, sm);
      }
      HIPCHK(c, hipGetLastError());
    }
    if (!tables_done && (rc = prepare_tables())) return rc;      // (a rank without scenes in this round still needs the frames)
    // ---- the round's decisions on the host: this rank's block, or (sharded) every rank's through one all-gather
    const float *h_dec;             // [ranks][Bmax x 24 floats]
    if (dist) {
      if ((rc = ensure(c, c->x_recv, (size_t)XW * n_back * sizeof(float)))) return rc;
      if ((rc = pl_exchange(c, MIND_XCHG_ALLGATHER, d_sel, c->x_recv.p, n_back * sizeof(float)))) return rc;
      if ((rc = pl_pin(c, 2, (size_t)XW * n_back * sizeof(float)))) return rc;
      HIPCHK(c, hipMemcpyAsync(c->pl_pin[2], c->x_recv.p, (size_t)XW * n_back * sizeof(float), hipMemcpyDeviceToHost, st));
    } else if (!h_mirror) {
      if ((rc = pl_pin(c, 2, n_back * sizeof(float)))) return rc;
      HIPCHK(c, hipMemcpyAsync(c->pl_pin[2], d_sel, n_back * sizeof(float), hipMemcpyDeviceToHost, st));
    }
    TR("round: glue launched, waiting for the decisions");
    HIPCHK(c, hipStreamSynchronize(st));
    TR("round: decisions on the host");
    h_dec = (const float *)c->pl_pin[2];
    // ---- create_nodes (scenario_tree.py:73-80): the kept modes scene by scene, visiting order within a scene
    for (int b = 0; b < B; ++b) {
      const int owner = dist ? pl_owner(B, XW, b) : 0;
      int olo, ohi;
      pl_block(B, owner, XW, olo, ohi);
      if (!dist) { olo = 0; }
      const float *h_sel = h_dec + (size_t)owner * n_back, *h_selp = h_sel + (size_t)Bmax * 6;
      const unsigned *h_hit = (const unsigned *)(h_selp + (size_t)Bmax * 6);
      const int bl = b - olo;
      for (int j = 0; j < AIME_K; ++j) {
        const int k = (int)h_sel[(size_t)bl * 6 + j];
        if (k < 0) continue;
        PlNode n;
        memset(&n, 0, sizeof(n));
        n.round = round; n.scene = b; n.mode = k; n.parent = batch[b].node; n.depth = nodes[n.parent].depth + 1;
        n.owner = owner; n.lscene = bl;
        n.prob = h_selp[(size_t)bl * 6 + j]; n.cur_t = batch[b].cur_t; n.end_t = batch[b].end_t;
        n.hit = (unsigned long long)h_hit[2 * ((size_t)bl * 6 + j)] | ((unsigned long long)h_hit[2 * ((size_t)bl * 6 + j) + 1] << 32);
        memcpy(n.tgt, batch[b].tgt, sizeof(n.tgt));
        const int idx = (int)nodes.size();
        nodes.push_back(n);
        for (size_t q = 0; q < leaves.size(); ++q)
          if (leaves[q] == n.parent) { leaves.erase(leaves.begin() + q); break; }
        leaves.push_back(idx);
      }
    }
    // ---- decide_branch (scenario_tree.py:82-100) over the leaves in insertion order
    std::vector<int> cand, todo;
    for (int li : leaves) {
      PlNode &n = nodes[li];
      if (n.branch) { n.branch = false; n.term = true; }
      else if (!n.end) {
        if (n.depth >= in->max_depth) n.term = true;
        else cand.push_back(li);
      }
    }
    for (int li : cand) {
      PlNode &n = nodes[li];
      if (li == 0) return fail(c, MIND_ESTATE, "unsupported: the root is a branching candidate");
      // get_branch_time (:592-611): first even t in (CUR_T, END_T) whose sigma ratio exceeds 9.  A node that was re-based before and
      // has CUR_T > 0 would index past its trimmed history in the reference: left to the host path.
      if (n.rebased && n.cur_t > 0) return fail(c, MIND_ESTATE, "unsupported: branch time of a trimmed node with CUR_T > 0");
      int t_b = n.end_t;
      for (int t = n.cur_t + 1 + (n.cur_t + 1) % 2; t < n.end_t; t += 2)
        if ((n.hit >> t) & 1ull) { t_b = t; break; }
      if (t_b < n.end_t) n.end_t = t_b;
      if (t_b < HZ) todo.push_back(li);
      else n.end = true;
    }
    if (todo.empty()) { ++round; break; }
    // ---- update_obser (:467-567) of the branching nodes: windows + predictor inputs of the next round, on the device.  Sharded: a rank
    //      re-bases the children of ITS scenes (the parent windows and the predicted rows are there); the branch set is in leaf order =
    //      parent-scene order, so a rank's children are one contiguous range [s0, s0 + Sm) of the next round's scenes
    const int S = (int)todo.size();
    for (int li : todo)
      if (nodes[li].round != round) return fail(c, MIND_ESTATE, "unsupported: a node of round %d is expanded again in round %d", nodes[li].round, round + 1);
    std::vector<int> cnt_r(XW, 0), s0_r(XW + 1, 0);
    for (int s = 0; s < S; ++s) {
      if (s > 0 && nodes[todo[s]].scene < nodes[todo[s - 1]].scene) return fail(c, MIND_ESTATE, "unsupported: branch set out of scene order");
      cnt_r[nodes[todo[s]].owner] += 1;
    }
    for (int r = 0; r < XW; ++r) s0_r[r + 1] = s0_r[r] + cnt_r[r];
    const int s0 = s0_r[XR], Sm = cnt_r[XR];
    const int nxt = cur_in < 0 ? 0 : cur_in ^ 1;
    const InOff q = in_off(S);
    if ((rc = ensure(c, c->pl_in[nxt], q.total * sizeof(float)))) return rc;
    const size_t n_wpos = (size_t)S * a * OBS * 2, n_wang = (size_t)S * a * OBS;
    if ((rc = ensure(c, c->pl_win[nxt], (2 * n_wpos + n_wang) * sizeof(float)))) return rc;
    float *w_pos = (float *)c->pl_win[nxt].p, *w_ang = w_pos + n_wpos, *w_vel = w_ang + n_wang;
    float *d_in = (float *)c->pl_in[nxt].p;
    if (Sm > 0) {
      int *hi_ = (int *)((char *)c->pl_pin[1] + tab_bytes);
      for (int s = 0; s < Sm; ++s) {
        const PlNode &n = nodes[todo[s0 + s]];
        hi_[s] = n.scene;                                  // parent window: the scene of this round the node was predicted from (global index)
        hi_[Sm + s] = n.lscene * a * AIME_K + n.mode;      // first row (agent 0) of the node's mode in this rank's d_world
        hi_[2 * Sm + s] = n.end_t - n.cur_t;               // steps kept
      }
      // (a small branch set: k_aime_windows reads its three ints per scene from the page-locked staging itself -- no copy in front of it)
      const bool tab_host = c->tab_host_max > 0 && (size_t)Sm * a <= (size_t)c->tab_host_max;
      if (!tab_host) HIPCHK(c, hipMemcpyAsync((char *)tabb.p + tab_bytes, hi_, 3 * (size_t)Sm * sizeof(int), hipMemcpyHostToDevice, st));
      const int *d_idx = tab_host ? (const int *)hi_ : (const int *)((const char *)tabb.p + tab_bytes);
      hipLaunchKernelGGL(k_aime_windows, dim3((unsigned)(Sm * a)), dim3(64), 0, st, prev_pos, prev_ang, prev_vel, (const float *)d_world, d_idx, d_idx + Sm,
                         d_idx + 2 * Sm, a, w_pos + (size_t)s0 * a * OBS * 2, w_ang + (size_t)s0 * a * OBS, w_vel + (size_t)s0 * a * OBS * 2, AIME_K,
                         d_in + q.cov + (size_t)s0 * a);
      RebaseArgs R;
      R.a = a; R.l = l; R.n_lane = P; R.pad_ones = 1;
      R.pos = w_pos + (size_t)s0 * a * OBS * 2; R.ang = w_ang + (size_t)s0 * a * OBS; R.vel = w_vel + (size_t)s0 * a * OBS * 2; R.types = droot + o_types; R.pad = nullptr;
      R.lane_ctrs0 = droot + o_lc; R.lane_vecs0 = droot + o_lv; R.tlane = droot + o_tl; R.tinfo = droot + o_ti;
      R.time_ahead = in->time_ahead; R.min_vel = in->min_vel; R.travel0 = -1.f;
      R.actors = d_in + q.actors + (size_t)s0 * a * 14 * 48; R.actor_ctrs = d_in + q.ctrs + (size_t)s0 * a * 2; R.actor_vecs = d_in + q.vecs + (size_t)s0 * a * 2;
      R.lane_ctrs = d_in + q.lc + (size_t)s0 * l * 2; R.lane_vecs = d_in + q.lv + (size_t)s0 * l * 2;
      R.tgt_nodes = d_in + q.tn + (size_t)s0 * 160; R.tgt_rpe = d_in + q.tr + (size_t)s0 * 20; R.frames = d_in + q.fr + (size_t)s0 * 28;
      hipLaunchKernelGGL(k_aime_rebase, dim3((unsigned)Sm, 1 + RB_FEAT_BLOCKS(a)), dim3(RB_THREADS), 5 * (size_t)a * sizeof(float), st, R);
      HIPCHK(c, hipGetLastError());
    }
    if (dist) {
      // ---- the next round's inputs + windows go where they are NEEDED: scene s of the next round was re-based by the rank that predicted its
      //      parent (its children are the contiguous range [s0_r, s0_r + cnt_r)) and is predicted -- and later branched from -- by the rank
      //      whose block [nlo, nhi) of the next round holds it.  On a full tree the two ranges coincide up to the block boundaries, so only
      //      the boundary scenes travel (all-to-all with per-pair sizes; round 4 handed every rank ALL scenes: 52 MB per cfg4 round,
      //      3.7 GB on the deepest stress tree).  What every rank does need of every scene is small: its frame (ROT / ORIG / TGT_PTS, 28 floats:
      //      the replicated tree's node records) and, after the root round, LaneNet's output -- one all-gather.
      struct Arr { float *base; size_t per; };
      const Arr arrs[11] = {{d_in + q.actors, (size_t)a * 14 * 48}, {d_in + q.ctrs, (size_t)a * 2}, {d_in + q.vecs, (size_t)a * 2}, {d_in + q.lc, (size_t)l * 2},
                            {d_in + q.lv, (size_t)l * 2}, {d_in + q.tn, 160}, {d_in + q.tr, 20}, {d_in + q.cov, (size_t)a},
                            {w_pos, (size_t)a * OBS * 2}, {w_ang, (size_t)a * OBS}, {w_vel, (size_t)a * OBS * 2}};
      float *fr_base = d_in + q.fr;
      int cmax = 0;
      for (int r = 0; r < XW; ++r) cmax = std::max(cmax, cnt_r[r]);
      // (1) all-gather: [LaneNet output (root round, from rank 0) | the frames of this rank's children, laid out for the largest child count]
      {
        const size_t n_hdr = round == 0 ? (size_t)l * 128 : 0;
        const size_t n_pay = (n_hdr + (size_t)cmax * 28 + 3) & ~(size_t)3;
        if ((rc = ensure(c, c->x_send, n_pay * sizeof(float)))) return rc;
        if ((rc = ensure(c, c->x_recv, (size_t)XW * n_pay * sizeof(float)))) return rc;
        float *snd = (float *)c->x_send.p, *rcv = (float *)c->x_recv.p;
        std::vector<CopySeg> segs;
        if (n_hdr && XR == 0) segs.push_back({(const float *)c->pl_lf.p, snd, (long long)n_hdr});
        if (Sm > 0) segs.push_back({fr_base + (size_t)s0 * 28, snd + n_hdr, (long long)Sm * 28});
        if ((rc = pl_copy_segs(c, segs))) return rc;
        if ((rc = pl_exchange(c, MIND_XCHG_ALLGATHER, snd, rcv, n_pay * sizeof(float)))) return rc;
        segs.clear();
        if (n_hdr && XR != 0) segs.push_back({rcv, (float *)c->pl_lf.p, (long long)n_hdr});
        for (int r = 0; r < XW; ++r) {
          if (r == XR || cnt_r[r] == 0) continue;
          segs.push_back({rcv + (size_t)r * n_pay + n_hdr, fr_base + (size_t)s0_r[r] * 28, (long long)cnt_r[r] * 28});
        }
        if ((rc = pl_copy_segs(c, segs))) return rc;
      }
      // (2) all-to-all: producer j -> consumer k, the scenes [max(s0_j, nlo_k), min(s0_j + cnt_j, nhi_k)); per pair the eleven array slices one
      //     behind the other.  Every rank computes the same table, so a round whose ranges coincide everywhere is skipped by all of them
      size_t per_scene = 0;
      for (int k = 0; k < 11; ++k) per_scene += arrs[k].per;
      auto isect = [&](int j, int k, int &i0, int &i1) {          // scenes rank j produced that rank k consumes
        int klo, khi;
        pl_block(S, k, XW, klo, khi);
        i0 = std::max(s0_r[j], klo); i1 = std::min(s0_r[j] + cnt_r[j], khi);
        if (i1 < i0) i1 = i0;
      };
      long long any = 0;
      std::vector<int64_t> tab(2 * (size_t)XW, 0);               // bytes this rank sends to / receives from every rank
      // (a forced group -- tests, overhead measurements: mind_set_exchange(force) -- also sends a rank's own scenes to itself through the
      // transport: the same values land where they are, and the all-to-all is exercised with real data even in a one-rank RCCL group)
      const bool self_too = c->xforce != 0;
      for (int j = 0; j < XW; ++j)
        for (int k = 0; k < XW; ++k) {
          if (j == k && !self_too) continue;
          int i0, i1;
          isect(j, k, i0, i1);
          any += i1 - i0;
          if (j == XR) tab[k] = (int64_t)(i1 - i0) * (int64_t)per_scene * (int64_t)sizeof(float);
          if (k == XR) tab[(size_t)XW + j] = (int64_t)(i1 - i0) * (int64_t)per_scene * (int64_t)sizeof(float);
        }
      if (any > 0) {
        size_t n_snd = 0, n_rcv = 0;
        for (int r = 0; r < XW; ++r) { n_snd += (size_t)tab[r] / sizeof(float); n_rcv += (size_t)tab[(size_t)XW + r] / sizeof(float); }
        if ((rc = ensure(c, c->x_send, (n_snd + 4) * sizeof(float)))) return rc;
        if ((rc = ensure(c, c->x_recv, (n_rcv + 4) * sizeof(float)))) return rc;
        float *snd = (float *)c->x_send.p, *rcv = (float *)c->x_recv.p;
        std::vector<CopySeg> segs;
        size_t o = 0;
        for (int k = 0; k < XW; ++k) {
          if (k == XR && !self_too) continue;
          int i0, i1;
          isect(XR, k, i0, i1);
          for (int e = 0; e < 11 && i1 > i0; ++e) {
            segs.push_back({arrs[e].base + arrs[e].per * (size_t)i0, snd + o, (long long)(arrs[e].per * (size_t)(i1 - i0))});
            o += arrs[e].per * (size_t)(i1 - i0);
          }
        }
        if ((rc = pl_copy_segs(c, segs))) return rc;
        if ((rc = pl_exchange_v(c, snd, rcv, tab.data()))) return rc;
        segs.clear();
        o = 0;
        for (int j = 0; j < XW; ++j) {
          if (j == XR && !self_too) continue;
          int i0, i1;
          isect(j, XR, i0, i1);
          for (int e = 0; e < 11 && i1 > i0; ++e) {
            segs.push_back({rcv + o, arrs[e].base + arrs[e].per * (size_t)i0, (long long)(arrs[e].per * (size_t)(i1 - i0))});
            o += arrs[e].per * (size_t)(i1 - i0);
          }
        }
        if ((rc = pl_copy_segs(c, segs))) return rc;
      }
    }
    // LaneNet's output repeated for this rank's scenes of the next round
    int nlo, nhi;
    pl_block(S, XR, XW, nlo, nhi);
    // (unsharded: the repeated lane features -- read by the token kernels, behind the predictor's side-stream work -- and the frames' read-back
    // go to the side stream; ActorNet follows the re-basing directly)
    hipStream_t rs = st;
    if (c->side && !dist) {
      if (!c->ev_root) HIPCHK(c, hipEventCreateWithFlags(&c->ev_root, hipEventDisableTiming));
      HIPCHK(c, hipEventRecord(c->ev_root, st));
      HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_root, 0));
      rs = c->side;
    }
    if (nhi - nlo > 1) {
      if ((rc = ensure(c, c->pl_lrep, (size_t)(nhi - nlo) * l * 128 * sizeof(float)))) return rc;
      const size_t n = (size_t)l * 128;
      hipLaunchKernelGGL(k_repeat_rows, dim3((unsigned)((n * (nhi - nlo) + 255) / 256)), dim3(256), 0, rs, (const float *)c->pl_lf.p, n, nhi - nlo, (float *)c->pl_lrep.p);
      HIPCHK(c, hipGetLastError());
    }
    if ((rc = pl_pin(c, 3, (size_t)S * 28 * sizeof(float)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pl_pin[3], d_in + q.fr, (size_t)S * 28 * sizeof(float), hipMemcpyDeviceToHost, rs));
    HIPCHK(c, hipEventRecord(c->ev_pl, rs));
    frames_pending = true;
    // next round's batch = the branch set in leaf order (scenario_tree.py:102-108)
    batch.assign(S, PlScene());
    for (int s = 0; s < S; ++s) {
      PlNode &n = nodes[todo[s]];
      n.branch = true; n.rebased = true;
      PlScene &sc = batch[s];
      sc.node = todo[s]; sc.prob = n.prob; sc.cur_t = n.end_t; sc.end_t = HZ;
    }
    prev_pos = w_pos; prev_ang = w_ang; prev_vel = w_vel;
    cov_last_dev = d_in + q.cov;
    cur_in = nxt;
  }
  // ---- get_scenario_tree, first step (:208-216): every node on a finished branch is labelled; their rows are packed by one kernel
  TR("rounds done");
  bool any_end = false;
  for (int li : leaves) any_end |= nodes[li].end;
  if (!any_end) return fail(c, MIND_ESTATE, "unsupported: no end node found in the scenario tree");
  for (int li : leaves) {
    if (!nodes[li].end) continue;
    for (int q = li; q > 0; q = nodes[q].parent) nodes[q].end = true;
  }
  const int N = (int)nodes.size() - 1;
  c->pl_nodes.assign(N, mind_aime_node());
  std::vector<AimeGather> jobs;
  std::vector<const float *> job_world;
  std::vector<int> job_of_block, agent_of_block;
  int64_t n_rows = 0;
  for (int i = 0; i < N; ++i) {
    const PlNode &n = nodes[i + 1];
    mind_aime_node &p = c->pl_nodes[i];
    p.round = n.round; p.scene = n.scene; p.mode = n.mode; p.parent = n.parent - 1; p.prob = n.prob; p.cur_t = n.cur_t; p.end_t = n.end_t;
    p.flags = (n.branch ? MIND_AIME_BRANCH : 0) | (n.end ? MIND_AIME_END : 0) | (n.term ? MIND_AIME_TERMINATE : 0);
    memcpy(p.tgt_pts, n.tgt, sizeof(p.tgt_pts));
    p.dur = 0; p.row_off = -1;
    if (n.end) {
      const int dur = n.end_t - n.cur_t;
      p.dur = dur; p.row_off = n_rows;
      if (dur > 0 && (!dist || n.owner == XR)) {            // (sharded: the rank that holds the node's predicted rows packs them)
        AimeGather J;
        J.row0 = n.lscene * a * AIME_K + n.mode; J.dur = dur; J.dst = (int)n_rows; J.a = a;
        for (int e = 0; e < a; ++e) { job_of_block.push_back((int)jobs.size()); agent_of_block.push_back(e); }
        jobs.push_back(J);
        job_world.push_back((const float *)c->pl_world[n.round].p);
      }
      n_rows += (int64_t)a * dur * 3;
    }
  }
  // ---- the cost trees of the finished branches, flattened as TrajectoryTreeOptimizer would (trajectory_tree.py:19-124 / get_scenario_tree
  //      :208-272): sibling-normalised probabilities in float64, LIFO depth-first creation order, every even step = one trajectory node
  auto &kids = c->pl_scr_kids;            // (scratch kept in the context: no allocations per plan)
  if (kids.size() < nodes.size()) kids.resize(nodes.size());
  for (size_t i = 0; i < nodes.size(); ++i) kids[i].clear();
  for (int i = 1; i < (int)nodes.size(); ++i) kids[nodes[i].parent].push_back(i);
  auto &pr = c->pl_scr_pr;
  auto &queue = c->pl_scr_i[0], &last = c->pl_scr_i[1], &stack = c->pl_scr_i[2];
  c->pl_tree_top.clear(); c->pl_tree_off.assign(1, 0); c->pl_flat_parent.clear(); c->pl_flat_prob.clear();
  std::vector<AimeFlat> fjobs;
  std::vector<const float *> fworld;
  std::vector<int> fjob_of_block, fagent_of_block;
  for (int top : kids[0]) {
    if (!nodes[top].end) continue;
    // probabilities: breadth-first renormalisation over the siblings that lie on finished branches
    // (float32 throughout: SCEN_PROB is a float32 scalar and the Python literals 0.0 / 1.0 it meets are weak scalars under numpy >= 2,
    // which is what the host path -- pinned against the reference's sibling probabilities in tests/golden/aime.npz -- computes with)
    pr.assign(nodes.size(), 0.f);
    pr[top] = 1.f;
    queue.assign(1, top);
    for (size_t qh = 0; qh < queue.size(); ++qh) {
      const int cur = queue[qh];
      float total = 0.f;
      for (int ch : kids[cur]) if (nodes[ch].end) total = total + nodes[ch].prob;
      for (int ch : kids[cur]) if (nodes[ch].end) { pr[ch] = nodes[ch].prob / total * pr[cur]; queue.push_back(ch); }
    }
    // flatten: stack pop() = the last child first; a node's trajectory nodes are chained, the first hangs off its parent's last
    const int base = c->pl_tree_off.back();
    int count = 0;
    last.assign(nodes.size(), -1); stack.assign(1, top);
    while (!stack.empty()) {
      const int q = stack.back();
      stack.pop_back();
      const PlNode &n = nodes[q];
      const int dur = n.end_t - n.cur_t, nn = (dur + 1) / 2;
      const int up = q == top ? -1 : last[n.parent];
      if (nn > 0) {
        for (int m = 0; m < nn; ++m) {
          c->pl_flat_parent.push_back(m == 0 ? up : count + m - 1);
          c->pl_flat_prob.push_back(pr[q]);
        }
        if (!dist || n.owner == XR) {
          AimeFlat J;
          J.row0 = n.lscene * a * AIME_K + n.mode; J.n = nn; J.dst = base + count; J.a = a;
          for (int e = 0; e < a; ++e) { fjob_of_block.push_back((int)fjobs.size()); fagent_of_block.push_back(e); }
          fjobs.push_back(J);
          fworld.push_back((const float *)c->pl_world[n.round].p);
        }
        count += nn;
        last[q] = count - 1;
      } else {
        last[q] = up;
      }
      for (int ch : kids[q]) if (nodes[ch].end) stack.push_back(ch);
    }
    c->pl_tree_top.push_back(top - 1);
    c->pl_tree_off.push_back(base + count);
  }
  const size_t Mtot = (size_t)c->pl_tree_off.back();
  c->pl_plan_agents = a;
  c->pl_rows_p = c->pl_fmean_p = c->pl_fcov_p = nullptr;
  c->pl_dev_fmean = c->pl_dev_fcov = nullptr;
  const bool want_solves = in->solve_cfg_full && !dist && !c->pl_tree_top.empty() && in->solve_x0 && in->solve_lane && in->solve_n_lane_pts >= 2;
  bool solves_tried = false;
  int solves_rc = MIND_OK;
  auto begin_solves = [&]() {
    // the contingency solves of the plan, begun by the plan itself: results stay in the library until mind_ilqr_finish_plan
    solves_tried = true;
    c->pl_sol_xs.resize(Mtot * 6); c->pl_sol_us.resize(Mtot * 2);
    c->pl_sol_stw.resize(c->pl_tree_top.size()); c->pl_sol_stf.resize(c->pl_tree_top.size());
    solves_rc = mind_ilqr_contingency_begin_plan(c, in->solve_cfg_warm, in->solve_cfg_full, in->solve_x0, in->solve_lane, in->solve_n_lane_pts,
                                                 in->solve_target_vel, c->pl_sol_xs.data(), c->pl_sol_us.data(), c->pl_sol_stw.data(), c->pl_sol_stf.data());
    c->il_finish_owned = solves_rc == MIND_OK;
  };
  // one upload (both job tables), the two gather kernels, one read-back (rows | flat means | flat covariances)
  const size_t n_flat = Mtot * a * 3;
  if (n_rows + (int64_t)n_flat > 0) {
    const size_t fJ = (fjobs.size() * sizeof(AimeFlat) + 15) & ~(size_t)15, fW = (fjobs.size() * sizeof(float *) + 15) & ~(size_t)15;
    const size_t fB = (fjob_of_block.size() * sizeof(int) + 15) & ~(size_t)15;
    const size_t gJ = (jobs.size() * sizeof(AimeGather) + 15) & ~(size_t)15, gW = (jobs.size() * sizeof(float *) + 15) & ~(size_t)15;
    const size_t gB = (job_of_block.size() * sizeof(int) + 15) & ~(size_t)15;
    const size_t o_g = fJ + fW + 2 * fB, n_tab = o_g + gJ + gW + 2 * gB;
    const size_t o_rows = (n_tab + 255) & ~(size_t)255;
    const size_t n_res = (size_t)n_rows + n_flat;
    if ((rc = ensure(c, c->pl_flat, o_rows + n_res * sizeof(float)))) return rc;
    if ((rc = pl_pin(c, 0, n_tab))) return rc;          // (the root upload of this plan completed rounds ago)
    char *h = (char *)c->pl_pin[0];
    if (!fjobs.empty()) {
      memcpy(h, fjobs.data(), fjobs.size() * sizeof(AimeFlat));
      memcpy(h + fJ, fworld.data(), fjobs.size() * sizeof(float *));
      memcpy(h + fJ + fW, fjob_of_block.data(), fjob_of_block.size() * sizeof(int));
      memcpy(h + fJ + fW + fB, fagent_of_block.data(), fagent_of_block.size() * sizeof(int));
    }
    if (!jobs.empty()) {
      memcpy(h + o_g, jobs.data(), jobs.size() * sizeof(AimeGather));
      memcpy(h + o_g + gJ, job_world.data(), jobs.size() * sizeof(float *));
      memcpy(h + o_g + gJ + gW, job_of_block.data(), job_of_block.size() * sizeof(int));
      memcpy(h + o_g + gJ + gW + gB, agent_of_block.data(), agent_of_block.size() * sizeof(int));
    }
    TR("end: trees flattened on the host");
    // (few jobs: the two packing kernels read their tables from the page-locked staging themselves)
    const bool tab_host = !dist && c->tab_host_max > 0 && fjob_of_block.size() + job_of_block.size() <= (size_t)c->tab_host_max;
    if (n_tab && !tab_host) HIPCHK(c, hipMemcpyAsync(c->pl_flat.p, h, n_tab, hipMemcpyHostToDevice, st));
    char *d = tab_host ? h : (char *)c->pl_flat.p;
    float *d_rows = (float *)((char *)c->pl_flat.p + o_rows), *d_fmean = d_rows + n_rows, *d_fcov = d_fmean + Mtot * a * 2;
    if (dist) HIPCHK(c, hipMemsetAsync(d_rows, 0, n_res * sizeof(float), st));      // every entry is written by exactly one rank: the sum completes it
    if (!fjobs.empty())
      hipLaunchKernelGGL(k_aime_flat, dim3((unsigned)fjob_of_block.size()), dim3(64), 0, st, (const AimeFlat *)d, (const int *)(d + fJ + fW),
                         (const int *)(d + fJ + fW + fB), (const float *const *)(d + fJ), d_fmean, d_fcov);
    auto gather = [&](hipStream_t gs) {
      if (!jobs.empty())
        hipLaunchKernelGGL(k_aime_gather, dim3((unsigned)job_of_block.size()), dim3(64), 0, gs, (const AimeGather *)(d + o_g),
                           (const int *)(d + o_g + gJ + gW), (const int *)(d + o_g + gJ + gW + gB), (const float *const *)(d + o_g + gJ), d_rows);
    };
    if (!want_solves) gather(st);
    HIPCHK(c, hipGetLastError());
    if (dist && (rc = pl_exchange(c, MIND_XCHG_ALLREDUCE, d_rows, d_rows, n_res * sizeof(float)))) return rc;
    c->pl_dev_fmean = d_fmean; c->pl_dev_fcov = d_fcov;
    if ((rc = pl_pin(c, 2, n_res * sizeof(float)))) return rc;
    float *hp = (float *)c->pl_pin[2];
    if (want_solves) {
      // k_ilqr right behind k_aime_flat: the cost trees' agent arrays stay where that kernel wrote them (ilqr_impl reads them on the device)
      // and the host builds the solver's tables while it runs.  What only the CALLER wants -- the finished branches' rows (k_aime_gather) and the
      // plan's read-back (rows | flat means | sigmas: the tree objects, the candidate evaluation) -- is queued on the copy stream AFTER the
      // solves have been launched: beside k_ilqr on the device, and behind its launch on the host (the five calls stood 10 us in front of it).
      if (!c->ev_rows) HIPCHK(c, hipEventCreateWithFlags(&c->ev_rows, hipEventDisableTiming));
      HIPCHK(c, hipEventRecord(c->ev_tab, st));          // (the job tables are up, the flat arrays written)
      TR("end: flat kernel queued");
      begin_solves();
      TR("end: solves begun");
      HIPCHK(c, hipStreamWaitEvent(c->pl_copy, c->ev_tab, 0));
      gather(c->pl_copy);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipMemcpyAsync(hp, d_rows, n_res * sizeof(float), hipMemcpyDeviceToHost, c->pl_copy));
      HIPCHK(c, hipEventRecord(c->ev_rows, c->pl_copy));
      HIPCHK(c, hipEventSynchronize(c->ev_rows));
      TR("end: read-back on the host");
    } else {
      HIPCHK(c, hipMemcpyAsync(hp, d_rows, n_res * sizeof(float), hipMemcpyDeviceToHost, st));
      HIPCHK(c, hipStreamSynchronize(st));
    }
    // the plan's rows and flattened cost trees are handed out where the read-back put them (page-locked slot 2: nothing touches it before
    // the context's next plan, the lifetime mind_aime_plan_out promises); the deep stress trees return 0.9 GB here
    c->pl_rows_p = hp; c->pl_fmean_p = hp + n_rows; c->pl_fcov_p = hp + n_rows + Mtot * a * 2;
  } else {
    HIPCHK(c, hipStreamSynchronize(st));
  }
  out->nodes = c->pl_nodes.data(); out->n_nodes = N;
  out->rows = c->pl_rows_p; out->n_row_floats = n_rows;
  out->n_expanded = n_expanded; out->n_rounds = round;
  out->root_flags = (nodes[0].branch ? MIND_AIME_BRANCH : 0) | (nodes[0].end ? MIND_AIME_END : 0) | (nodes[0].term ? MIND_AIME_TERMINATE : 0);
  if (c->profiling && (rc = mind_pair_events_resolve(c, &pair_ms))) return rc;      // (the plan's kernels have completed: every exit above waited for the read-back behind them)
  out->pair_ms = pair_ms; out->pair_launches = pair_launches;
  out->n_trees = (int)c->pl_tree_top.size(); out->tree_top = c->pl_tree_top.data(); out->tree_off = c->pl_tree_off.data();
  out->flat_parent = c->pl_flat_parent.data(); out->flat_prob = c->pl_flat_prob.data();
  out->flat_mean = c->pl_fmean_p; out->flat_cov = c->pl_fcov_p;
  // ---- the contingency solves of the plan, begun here when the caller handed their inputs in (no host round trip between the plan's
  //      read-back and k_ilqr); results stay in the library until mind_ilqr_finish_plan
  if (want_solves) {
    if (!solves_tried) begin_solves();
    out->solves_begun = solves_rc == MIND_OK ? 1 : 0;      // (a failed begin is not the plan's failure: the caller then solves the usual way)
  }
  TR("end: result handed out");
  return MIND_OK;
}

extern "C" int mind_ilqr_finish_plan(mind_ctx *c, int n_nodes, int n_trees, double *xs, double *us, mind_ilqr_stats *stats_warm, mind_ilqr_stats *stats_full) {
  if (!c || !xs || !us || !stats_full) return MIND_EINVAL;
  if (!c->il_finish || !c->il_finish_owned) return fail(c, MIND_ESTATE, "mind_ilqr_finish_plan: no plan-begun tree-iLQR call is pending on this context");
  const int rc = mind_ilqr_finish(c);
  if (rc) return rc;
  // the caller sized xs / us / stats_* from ITS plan's tree table: refuse to copy another plan's results into them
  if ((size_t)n_nodes * 6 != c->pl_sol_xs.size() || (size_t)n_trees != c->pl_sol_stf.size() || (size_t)n_trees + 1 != c->pl_tree_off.size() ||
      c->pl_tree_off[(size_t)n_trees] != n_nodes)
    return fail(c, MIND_EINVAL, "mind_ilqr_finish_plan: the caller expects %d nodes in %d trees, the pending solves hold %zu in %zu", n_nodes, n_trees,
                c->pl_sol_xs.size() / 6, c->pl_sol_stf.size());
  memcpy(xs, c->pl_sol_xs.data(), c->pl_sol_xs.size() * sizeof(double));
  memcpy(us, c->pl_sol_us.data(), c->pl_sol_us.size() * sizeof(double));
  if (stats_warm) memcpy(stats_warm, c->pl_sol_stw.data(), c->pl_sol_stw.size() * sizeof(mind_ilqr_stats));
  memcpy(stats_full, c->pl_sol_stf.data(), c->pl_sol_stf.size() * sizeof(mind_ilqr_stats));
  return MIND_OK;
}


// ---- mind_aime_plan in two halves (a host thread that plans several scenes, one context each: mind_amd/pipelined.py)
extern "C" int mind_aime_plan_begin(mind_ctx *c, const mind_aime_plan_in *in) {
  if (!c || !in) return MIND_EINVAL;
  // a plan that is still RUNNING must be collected first; one that finished and was never collected (its caller gave it up) is dropped
  if (c->pa_state.load(std::memory_order_acquire) == 1) return fail(c, MIND_ESTATE, "mind_aime_plan_begin: a plan is running on this context");
  if (c->pa_thread.joinable()) c->pa_thread.join();
  c->pa_in = *in;
  memset(&c->pa_out, 0, sizeof(c->pa_out));
  c->pa_state.store(1, std::memory_order_release);
  try {
    c->pa_thread = std::thread([c] {
      c->pa_rc = mind_aime_plan(c, &c->pa_in, &c->pa_out);
      c->pa_state.store(2, std::memory_order_release);
    });
  } catch (...) {
    c->pa_state.store(0, std::memory_order_release);
    return fail(c, MIND_ESTATE, "mind_aime_plan_begin: no thread");
  }
  return MIND_OK;
}

extern "C" int mind_aime_plan_poll(mind_ctx *c) {
  if (!c) return MIND_EINVAL;
  return c->pa_state.load(std::memory_order_acquire) == 1 ? 1 : 0;
}

extern "C" int mind_aime_plan_finish(mind_ctx *c, mind_aime_plan_out *out) {
  if (!c || !out) return MIND_EINVAL;
  if (c->pa_state.load(std::memory_order_acquire) == 0) return fail(c, MIND_ESTATE, "mind_aime_plan_finish: no plan was begun");
  if (c->pa_thread.joinable()) c->pa_thread.join();
  *out = c->pa_out;
  c->pa_state.store(0, std::memory_order_release);
  return c->pa_rc;
}

extern "C" int mind_ctx_busy(mind_ctx *c) {
  if (!c) return MIND_EINVAL;
  if (c->pa_state.load(std::memory_order_acquire) == 1) return 1;
  (void)hipSetDevice(c->device);
  const hipError_t e = hipStreamQuery(c->stream);
  if (e == hipErrorNotReady) return 1;
  if (e != hipSuccess) return fail(c, MIND_EHIP, "mind_ctx_busy: %s", hipGetErrorString(e));
  return 0;
}
