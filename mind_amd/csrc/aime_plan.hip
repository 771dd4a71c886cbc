// mind_aime_plan: ScenarioTreeGenerator.branch_aime (planners/mind/scenario_tree.py:38-58) in one call.  Included by mind_hip.hip
// (uses mind_ctx, ensure / fail / HIPCHK, mind_predict_batch and the kernels of aime_kernels.hip).
//
// Per round the device runs  predictor -> k_aime_world -> k_aime_select -> k_aime_branch  and the host reads ONE small buffer (kept
// modes, path probabilities, branch-time bits: 72 B per scene); create_nodes / decide_branch (scenario_tree.py:73-100) are restated
// here over plain arrays; the branching nodes' observations are re-based where they are (k_aime_windows from k_aime_world's rows
// and the parents' windows, k_aime_rebase) and the next predictor call is queued before the host looks at anything else.  Every
// small table goes through page-locked staging, so no copy waits for the stream to drain.  At the end one gather kernel packs the
// rows get_scenario_tree (scenario_tree.py:208-272) attaches to the nodes of finished branches.
namespace {

struct PlScene {            // an observation pushed through the predictor: the root or a re-based branch node
  int node;                 // internal tree node it belongs to (0 = root)
  float prob;
  int cur_t, end_t;
  float rot[4], orig[2], tgt[22];
};

struct PlNode {
  int round, scene, mode, parent, depth;
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
  const size_t want = bytes + bytes / 2 + 4096;
  if (hipHostMalloc(&c->pl_pin[which], want, hipHostMallocDefault) != hipSuccess)
    return fail(c, MIND_ENOMEM, "hipHostMalloc(%zu) failed", want);
  c->pl_pin_cap[which] = want;
  return MIND_OK;
}

}  // namespace

extern "C" int mind_aime_plan(mind_ctx *c, const mind_aime_plan_in *in, mind_aime_plan_out *out) {
  if (!c || !in || !out) return MIND_EINVAL;
  if (!c->have_weights) return fail(c, MIND_ESTATE, "weights not loaded");
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
    HIPCHK(c, hipMemcpyAsync((float *)c->pl_root.p + o_types, h + o_types, (n_root - o_types) * sizeof(float), hipMemcpyHostToDevice, st));
    float *d = (float *)c->pl_root.p;
    RebaseArgs R;
    R.a = a; R.l = 0; R.n_lane = P; R.pad_ones = 0;
    R.pos = d + o_rpos; R.ang = d + o_rang; R.vel = d + o_rvel; R.types = d + o_types; R.pad = d + o_rpad;
    R.lane_ctrs0 = nullptr; R.lane_vecs0 = nullptr; R.tlane = d + o_tl; R.tinfo = d + o_ti;
    R.time_ahead = in->time_ahead; R.min_vel = in->min_vel; R.travel0 = in->travel0;
    R.actors = d + o_actors; R.actor_ctrs = d + o_ctrs; R.actor_vecs = d + o_vecs; R.lane_ctrs = nullptr; R.lane_vecs = nullptr;
    R.tgt_nodes = d + o_tn; R.tgt_rpe = d + o_tr; R.frames = d + o_fr;
    hipLaunchKernelGGL(k_aime_rebase, dim3(1, 1 + RB_FEAT_BLOCKS(a)), dim3(RB_THREADS), 5 * (size_t)a * sizeof(float), st, R);
    hipLaunchKernelGGL(k_aime_root_lanes, dim3(l), dim3(64), 0, st, (const double *)(d + o_lpts), (const int *)(d + o_lfl), (const float *)(d + o_fr),
                       d + o_lc, d + o_lv, d + o_lanes);
    hipLaunchKernelGGL(k_aime_root_hist, dim3(a), dim3(64), 0, st, (const float *)(d + o_rpos), (const float *)(d + o_rang), (const float *)(d + o_rvel),
                       (const float *)(d + o_fr), (const float *)(d + o_ctrs), (const float *)(d + o_vecs), d + o_wpos, d + o_wang, d + o_wvel, d + o_cov);
    HIPCHK(c, hipGetLastError());
    if ((rc = pl_pin(c, 3, 28 * sizeof(float)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pl_pin[3], d + o_fr, 28 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipEventRecord(c->ev_pl, st));
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

  // layout of one re-based input set (floats): actors | actor_ctrs | actor_vecs | lane_ctrs | lane_vecs | tgt_nodes | tgt_rpe | frames | cov_last
  struct InOff { size_t actors, ctrs, vecs, lc, lv, tn, tr, fr, cov, total; };
  auto in_off = [&](size_t S) {
    InOff q; size_t p = 0;
    auto tk = [&](size_t n) { const size_t r = p; p += (n + 3) & ~(size_t)3; return r; };
    q.actors = tk(S * a * 14 * 48); q.ctrs = tk(S * a * 2); q.vecs = tk(S * a * 2); q.lc = tk(S * l * 2); q.lv = tk(S * l * 2);
    q.tn = tk(S * 160); q.tr = tk(S * 20); q.fr = tk(S * 28); q.cov = tk(S * a); q.total = p;
    return q;
  };

  for (;; ++round) {
    const int B = (int)batch.size(), A = B * a;
    if (round >= in->max_rounds) return fail(c, MIND_ESTATE, "unsupported: more than %d AIME rounds", in->max_rounds);
    out->round_scenes[round] = B;
    // ---- predictor on the round's scenes
    std::vector<int32_t> ao(B + 1), lo(B + 1);
    for (int b = 0; b <= B; ++b) { ao[b] = a * b; lo[b] = l * b; }
    if ((rc = ensure(c, c->pl_pred, ((size_t)B * 6 + (size_t)A * 6 * T * 7) * sizeof(float)))) return rc;
    float *d_cls = (float *)c->pl_pred.p, *d_reg = d_cls + (((size_t)B * 6 + 3) & ~(size_t)3), *d_vel = d_reg + (size_t)A * 6 * T * 5;
    if ((rc = ensure(c, c->pl_pred, ((size_t)(d_vel - d_cls) + (size_t)A * 6 * T * 2) * sizeof(float)))) return rc;
    d_cls = (float *)c->pl_pred.p; d_reg = d_cls + (((size_t)B * 6 + 3) & ~(size_t)3); d_vel = d_reg + (size_t)A * 6 * T * 5;
    mind_scene_batch sb;
    memset(&sb, 0, sizeof(sb));
    mind_pred_out po;
    memset(&po, 0, sizeof(po));
    sb.n_scenes = B; sb.actor_off = ao.data(); sb.lane_off = lo.data();
    const float *d_ctrs, *d_vecs;
    if (cur_in < 0) {
      sb.actors = droot + o_actors; sb.lanes = droot + o_lanes; sb.actor_ctrs = d_ctrs = droot + o_ctrs; sb.actor_vecs = d_vecs = droot + o_vecs;
      sb.lane_ctrs = droot + o_lc; sb.lane_vecs = droot + o_lv; sb.tgt_nodes = droot + o_tn; sb.tgt_rpe = droot + o_tr;
      po.lane_feat = (float *)c->pl_lf.p;
    } else {
      const InOff q = in_off(B);
      const float *d = (const float *)c->pl_in[cur_in].p;
      sb.actors = d + q.actors; sb.actor_ctrs = d_ctrs = d + q.ctrs; sb.actor_vecs = d_vecs = d + q.vecs; sb.lane_ctrs = d + q.lc; sb.lane_vecs = d + q.lv;
      sb.tgt_nodes = d + q.tn; sb.tgt_rpe = d + q.tr;
      sb.lane_feat = B == 1 ? (const float *)c->pl_lf.p : (const float *)c->pl_lrep.p;
    }
    po.cls = d_cls; po.reg = d_reg; po.vel = d_vel;
    if ((rc = mind_predict_batch(c, &sb, &po))) return rc;
    n_expanded += B;
    if (c->profiling) { pair_ms += c->pair_ms; pair_launches += c->n_pair_launch; }
    if (in->script_cls) {
      // scripted modes (benchmark hook): the forward above was the timed work, its outputs are replaced scene by scene
      const size_t nr = (size_t)a * 6 * T * 5, nv = (size_t)a * 6 * T * 2;
      hipLaunchKernelGGL(k_repeat_rows, dim3((unsigned)((6 * (size_t)B + 255) / 256)), dim3(256), 0, st, in->script_cls, (size_t)6, B, d_cls);
      hipLaunchKernelGGL(k_repeat_rows, dim3((unsigned)((nr * B + 255) / 256)), dim3(256), 0, st, in->script_reg, nr, B, d_reg);
      hipLaunchKernelGGL(k_repeat_rows, dim3((unsigned)((nv * B + 255) / 256)), dim3(256), 0, st, in->script_vel, nv, B, d_vel);
    }
    // ---- the frames of the re-based scenes (queued behind k_aime_rebase, before this round's predictor): ROT, ORIG, TGT_PTS
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
    // ---- prune_merge arithmetic + decisions + branch-time bits on the device
    const size_t bS = ((size_t)B * sizeof(AimeScene) + 15) & ~(size_t)15, bI = ((size_t)A * sizeof(int) + 15) & ~(size_t)15;
    const size_t bP = ((size_t)B * sizeof(float) + 15) & ~(size_t)15;
    const size_t tab_bytes = bS + bI + bP;
    // (two table buffers, by round parity: this round's tables are uploaded on the copy stream while the predictor runs -- the stream
    // may still hold the previous round's windows kernel, which reads the index lists behind that round's tables)
    DevBuf &tabb = c->pl_tab[round & 1];
    if ((rc = ensure(c, tabb, tab_bytes + 3 * (size_t)6 * B * sizeof(int) + 64))) return rc;
    if ((rc = pl_pin(c, 1, tab_bytes + 3 * (size_t)6 * B * sizeof(int) + 64))) return rc;
    {
      char *h = (char *)c->pl_pin[1];
      AimeScene *hs = (AimeScene *)h;
      int *as = (int *)(h + bS);
      float *sp = (float *)(h + bS + bI);
      for (int b = 0; b < B; ++b) {
        AimeScene &S = hs[b];
        const PlScene &q = batch[b];
        S.a0 = a * b; S.a1 = a * (b + 1); S.last = HZ - 1;     /* seq_len - 1 - history length */ S.cmp = q.cur_t == 0 ? 1 : q.cur_t; S.pad2 = 0.f;
        S.r00 = q.rot[0]; S.r01 = q.rot[1]; S.r10 = q.rot[2]; S.r11 = q.rot[3];
        S.ox = q.orig[0]; S.oy = q.orig[1];
        S.theta_g = atan2f(S.r10, S.r00);
        for (int i = S.a0; i < S.a1; ++i) as[i] = b;
        sp[b] = q.prob;
      }
      if (c->pl_tab_side) {
        HIPCHK(c, hipMemcpyAsync(tabb.p, h, tab_bytes, hipMemcpyHostToDevice, c->pl_copy));
        HIPCHK(c, hipEventRecord(c->ev_tab, c->pl_copy));
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_tab, 0));
      } else {
        HIPCHK(c, hipMemcpyAsync(tabb.p, h, tab_bytes, hipMemcpyHostToDevice, st));
      }
    }
    const char *dtab = (const char *)tabb.p;
    if ((int)c->pl_world.size() <= round) c->pl_world.resize(round + 1);
    if ((rc = ensure(c, c->pl_world[round], (size_t)A * 6 * T * 6 * sizeof(float)))) return rc;
    float *d_world = (float *)c->pl_world[round].p;
    // small outputs: topo [A,6] | ego_end [B,6,4] | sel [B,6] | sel_prob [B,6] | hit [B,6,2] (the last three are read back)
    const size_t n_topo = ((size_t)A * 6 + 3) & ~(size_t)3, n_ego = (size_t)B * 24, n_back = (size_t)B * 6 * 4;
    if ((rc = ensure(c, c->pl_small, (n_topo + n_ego + n_back) * sizeof(float)))) return rc;
    float *d_topo = (float *)c->pl_small.p, *d_ego = d_topo + n_topo, *d_sel = d_ego + n_ego, *d_selp = d_sel + (size_t)B * 6;
    unsigned *d_hit = (unsigned *)(d_selp + (size_t)B * 6);
    hipLaunchKernelGGL(k_aime_world, dim3(A * AIME_K), dim3(64), 0, st, (const AimeScene *)dtab, (const int *)(dtab + bS), d_reg, d_vel, d_ctrs, d_vecs,
                       cov_last_dev, d_world, d_topo, d_ego, droot + o_tl, P);
    hipLaunchKernelGGL(k_aime_select, dim3(B), dim3(64), 0, st, (const AimeScene *)dtab, d_cls, (const float *)(dtab + bS + bI), d_topo, d_ego, 1,
                       in->dist_thres, d_sel, d_selp);
    hipLaunchKernelGGL(k_aime_branch, dim3(B * AIME_K), dim3(64), 0, st, (const AimeScene *)dtab, d_sel, d_world, d_hit);
    HIPCHK(c, hipGetLastError());
    if ((rc = pl_pin(c, 2, n_back * sizeof(float)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pl_pin[2], d_sel, n_back * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const float *h_sel = (const float *)c->pl_pin[2], *h_selp = h_sel + (size_t)B * 6;
    const unsigned *h_hit = (const unsigned *)(h_selp + (size_t)B * 6);
    // ---- create_nodes (scenario_tree.py:73-80): the kept modes scene by scene, visiting order within a scene
    for (int b = 0; b < B; ++b)
      for (int j = 0; j < AIME_K; ++j) {
        const int k = (int)h_sel[(size_t)b * 6 + j];
        if (k < 0) continue;
        PlNode n;
        memset(&n, 0, sizeof(n));
        n.round = round; n.scene = b; n.mode = k; n.parent = batch[b].node; n.depth = nodes[n.parent].depth + 1;
        n.prob = h_selp[(size_t)b * 6 + j]; n.cur_t = batch[b].cur_t; n.end_t = batch[b].end_t;
        n.hit = (unsigned long long)h_hit[2 * ((size_t)b * 6 + j)] | ((unsigned long long)h_hit[2 * ((size_t)b * 6 + j) + 1] << 32);
        memcpy(n.tgt, batch[b].tgt, sizeof(n.tgt));
        const int idx = (int)nodes.size();
        nodes.push_back(n);
        for (size_t q = 0; q < leaves.size(); ++q)
          if (leaves[q] == n.parent) { leaves.erase(leaves.begin() + q); break; }
        leaves.push_back(idx);
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
    // ---- update_obser (:467-567) of the branching nodes: windows + predictor inputs of the next round, on the device
    const int S = (int)todo.size();
    for (int li : todo)
      if (nodes[li].round != round) return fail(c, MIND_ESTATE, "unsupported: a node of round %d is expanded again in round %d", nodes[li].round, round + 1);
    {
      int *hi = (int *)((char *)c->pl_pin[1] + tab_bytes);
      for (int s = 0; s < S; ++s) {
        const PlNode &n = nodes[todo[s]];
        hi[s] = n.scene;                                  // parent window: the scene of this round the node was predicted from
        hi[S + s] = n.scene * a * AIME_K + n.mode;        // first row (agent 0) of the node's mode in d_world
        hi[2 * S + s] = n.end_t - n.cur_t;                // steps kept
      }
      HIPCHK(c, hipMemcpyAsync((char *)tabb.p + tab_bytes, hi, 3 * (size_t)S * sizeof(int), hipMemcpyHostToDevice, st));
    }
    const int *d_idx = (const int *)((const char *)tabb.p + tab_bytes);
    const int nxt = cur_in < 0 ? 0 : cur_in ^ 1;
    const InOff q = in_off(S);
    if ((rc = ensure(c, c->pl_in[nxt], q.total * sizeof(float)))) return rc;
    const size_t n_wpos = (size_t)S * a * OBS * 2, n_wang = (size_t)S * a * OBS;
    if ((rc = ensure(c, c->pl_win[nxt], (2 * n_wpos + n_wang) * sizeof(float)))) return rc;
    float *w_pos = (float *)c->pl_win[nxt].p, *w_ang = w_pos + n_wpos, *w_vel = w_ang + n_wang;
    float *d_in = (float *)c->pl_in[nxt].p;
    hipLaunchKernelGGL(k_aime_windows, dim3((unsigned)(S * a)), dim3(64), 0, st, prev_pos, prev_ang, prev_vel, (const float *)d_world, d_idx, d_idx + S,
                       d_idx + 2 * S, a, w_pos, w_ang, w_vel, AIME_K, d_in + q.cov);
    RebaseArgs R;
    R.a = a; R.l = l; R.n_lane = P; R.pad_ones = 1;
    R.pos = w_pos; R.ang = w_ang; R.vel = w_vel; R.types = droot + o_types; R.pad = nullptr;
    R.lane_ctrs0 = droot + o_lc; R.lane_vecs0 = droot + o_lv; R.tlane = droot + o_tl; R.tinfo = droot + o_ti;
    R.time_ahead = in->time_ahead; R.min_vel = in->min_vel; R.travel0 = -1.f;
    R.actors = d_in + q.actors; R.actor_ctrs = d_in + q.ctrs; R.actor_vecs = d_in + q.vecs; R.lane_ctrs = d_in + q.lc; R.lane_vecs = d_in + q.lv;
    R.tgt_nodes = d_in + q.tn; R.tgt_rpe = d_in + q.tr; R.frames = d_in + q.fr;
    hipLaunchKernelGGL(k_aime_rebase, dim3((unsigned)S, 1 + RB_FEAT_BLOCKS(a)), dim3(RB_THREADS), 5 * (size_t)a * sizeof(float), st, R);
    if (S > 1) {
      if ((rc = ensure(c, c->pl_lrep, (size_t)S * l * 128 * sizeof(float)))) return rc;
      const size_t n = (size_t)l * 128;
      hipLaunchKernelGGL(k_repeat_rows, dim3((unsigned)((n * S + 255) / 256)), dim3(256), 0, st, (const float *)c->pl_lf.p, n, S, (float *)c->pl_lrep.p);
    }
    HIPCHK(c, hipGetLastError());
    if ((rc = pl_pin(c, 3, (size_t)S * 28 * sizeof(float)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pl_pin[3], d_in + q.fr, (size_t)S * 28 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipEventRecord(c->ev_pl, st));
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
      if (dur > 0) {
        AimeGather J;
        J.row0 = n.scene * a * AIME_K + n.mode; J.dur = dur; J.dst = (int)n_rows; J.a = a;
        for (int e = 0; e < a; ++e) { job_of_block.push_back((int)jobs.size()); agent_of_block.push_back(e); }
        jobs.push_back(J);
        job_world.push_back((const float *)c->pl_world[n.round].p);
      }
      n_rows += (int64_t)a * dur * 3;
    }
  }
  // ---- the cost trees of the finished branches, flattened as TrajectoryTreeOptimizer would (trajectory_tree.py:19-124 / get_scenario_tree
  //      :208-272): sibling-normalised probabilities in float64, LIFO depth-first creation order, every even step = one trajectory node
  std::vector<std::vector<int>> kids(nodes.size());
  for (int i = 1; i < (int)nodes.size(); ++i) kids[nodes[i].parent].push_back(i);
  c->pl_tree_top.clear(); c->pl_tree_off.assign(1, 0); c->pl_flat_parent.clear(); c->pl_flat_prob.clear();
  std::vector<AimeFlat> fjobs;
  std::vector<const float *> fworld;
  std::vector<int> fjob_of_block, fagent_of_block;
  for (int top : kids[0]) {
    if (!nodes[top].end) continue;
    // probabilities: breadth-first renormalisation over the siblings that lie on finished branches
    // (float32 throughout: SCEN_PROB is a float32 scalar and the Python literals 0.0 / 1.0 it meets are weak scalars under numpy >= 2,
    // which is what the host path -- pinned against the reference's sibling probabilities in tests/golden/aime.npz -- computes with)
    std::vector<float> pr(nodes.size(), 0.f);
    pr[top] = 1.f;
    std::vector<int> queue(1, top);
    for (size_t qh = 0; qh < queue.size(); ++qh) {
      const int cur = queue[qh];
      float total = 0.f;
      for (int ch : kids[cur]) if (nodes[ch].end) total = total + nodes[ch].prob;
      for (int ch : kids[cur]) if (nodes[ch].end) { pr[ch] = nodes[ch].prob / total * pr[cur]; queue.push_back(ch); }
    }
    // flatten: stack pop() = the last child first; a node's trajectory nodes are chained, the first hangs off its parent's last
    const int base = c->pl_tree_off.back();
    int count = 0;
    std::vector<int> last(nodes.size(), -1), stack(1, top);
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
        AimeFlat J;
        J.row0 = n.scene * a * AIME_K + n.mode; J.n = nn; J.dst = base + count; J.a = a;
        for (int e = 0; e < a; ++e) { fjob_of_block.push_back((int)fjobs.size()); fagent_of_block.push_back(e); }
        fjobs.push_back(J);
        fworld.push_back((const float *)c->pl_world[n.round].p);
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
  c->pl_flat_mean.resize(Mtot * a * 2); c->pl_flat_cov.resize(Mtot * a);
  c->pl_rows_host.resize((size_t)n_rows);
  // one upload (both job tables), the two gather kernels, one read-back (rows | flat means | flat covariances)
  const size_t n_flat = fjobs.empty() ? 0 : Mtot * a * 3;
  if (!jobs.empty() || !fjobs.empty()) {
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
    HIPCHK(c, hipMemcpyAsync(c->pl_flat.p, h, n_tab, hipMemcpyHostToDevice, st));
    char *d = (char *)c->pl_flat.p;
    float *d_rows = (float *)(d + o_rows), *d_fmean = d_rows + n_rows, *d_fcov = d_fmean + Mtot * a * 2;
    if (!fjobs.empty())
      hipLaunchKernelGGL(k_aime_flat, dim3((unsigned)fjob_of_block.size()), dim3(64), 0, st, (const AimeFlat *)d, (const int *)(d + fJ + fW),
                         (const int *)(d + fJ + fW + fB), (const float *const *)(d + fJ), d_fmean, d_fcov);
    if (!jobs.empty())
      hipLaunchKernelGGL(k_aime_gather, dim3((unsigned)job_of_block.size()), dim3(64), 0, st, (const AimeGather *)(d + o_g),
                         (const int *)(d + o_g + gJ + gW), (const int *)(d + o_g + gJ + gW + gB), (const float *const *)(d + o_g + gJ), d_rows);
    HIPCHK(c, hipGetLastError());
    if ((rc = pl_pin(c, 2, n_res * sizeof(float)))) return rc;
    float *hp = (float *)c->pl_pin[2];
    HIPCHK(c, hipMemcpyAsync(hp, d_rows, n_res * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    memcpy(c->pl_rows_host.data(), hp, (size_t)n_rows * sizeof(float));
    if (n_flat) {
      memcpy(c->pl_flat_mean.data(), hp + n_rows, Mtot * a * 2 * sizeof(float));
      memcpy(c->pl_flat_cov.data(), hp + n_rows + Mtot * a * 2, Mtot * a * sizeof(float));
    }
  } else {
    HIPCHK(c, hipStreamSynchronize(st));
  }
  out->nodes = c->pl_nodes.data(); out->n_nodes = N;
  out->rows = c->pl_rows_host.data(); out->n_row_floats = n_rows;
  out->n_expanded = n_expanded; out->n_rounds = round;
  out->root_flags = (nodes[0].branch ? MIND_AIME_BRANCH : 0) | (nodes[0].end ? MIND_AIME_END : 0) | (nodes[0].term ? MIND_AIME_TERMINATE : 0);
  out->pair_ms = pair_ms; out->pair_launches = pair_launches;
  out->n_trees = (int)c->pl_tree_top.size(); out->tree_top = c->pl_tree_top.data(); out->tree_off = c->pl_tree_off.data();
  out->flat_parent = c->pl_flat_parent.data(); out->flat_prob = c->pl_flat_prob.data();
  out->flat_mean = c->pl_flat_mean.data(); out->flat_cov = c->pl_flat_cov.data();
  return MIND_OK;
}
