// mind_loop_*: the closed loop of one scene in native code (include/mind_hip.h, "The closed loop of one scene behind ONE call").
// Host code only: the device work is what mind_aime_plan / the plan-begun contingency solves already do; this file is the part of the
// cycle the interpreter used to run between them (reference: simulator.py:51-107, agent.py:255-331, planner.py:50-145, utils.py:245-342,
// kinematics.py:22-36; this repo's Python form of the same steps: mind_amd/closed_loop.py, planners/mind/planner.py).
// Included at the end of mind_hip.hip (needs mind_ctx, fail, mind_aime_plan, mind_ilqr_finish, np_pairwise / eval_nodes).
#pragma once
#pragma STDC FP_CONTRACT OFF
#include <chrono>

namespace {

constexpr int LOOP_OBS = 50;        // MINDPlanner.obs_len (planner.py:14)
constexpr int LOOP_ROW = 7;         // observed, x, y, heading, vx, vy, timestep

struct LoopTrack {
  int track = 0;                    // index into the world's tracks
  int count = 0, head = 0;          // rows in the window, position of the oldest
  double rows[LOOP_OBS][LOOP_ROW];
  const double *last() const { return rows[(head + count - 1) % LOOP_OBS]; }
  void push(const double *r) {
    double *dst;
    if (count < LOOP_OBS) dst = rows[(head + count++) % LOOP_OBS];
    else { dst = rows[head]; head = (head + 1) % LOOP_OBS; }            // object_states.pop(0)
    memcpy(dst, r, LOOP_ROW * sizeof(double));
  }
};

}  // namespace

struct mind_loop {
  mind_ctx *c = nullptr;
  mind_loop_desc d;
  // copies of the caller's tables
  std::vector<double> ego_state, ego_obs, exo_obs, lane_pts, solve_lane, eval_lane_f64;
  std::vector<uint8_t> exo_valid;
  std::vector<int32_t> timestep, type_slot, lane_flags;
  std::vector<float> target_lane, target_lane_info, eval_lane_f32, ego_trig32;
  mind_ilqr_cfg cfg_warm, cfg_full;
  // simulator state (ClosedLoopSim)
  double sim_time = 0.0, last_trigger = -1.0;
  bool have_trigger = false, enabled = false;
  bool state_recorded = false;      // the plant state is the recorded one of row state_row (taken over in this step, not propagated yet)
  int state_row = 0;
  bool tan_valid = false;
  double tan_arg = 0.0, tan_val = 0.0;
  double state[4] = {0, 0, 0, 0}, ctrl[2] = {0, 0};
  long long n_steps = 0, n_plans = 0, ep_steps = 0;
  // observation windows (MINDPlanner.agent_obs): tracks in first-appearance order, AV first
  std::vector<LoopTrack> obs;
  std::vector<int> slot_of_track;       // world track -> index in obs, -1 = not seen yet
  // per-plan scratch + the last plan
  std::vector<double> raw, eval_st, eval_ct, costs, per;
  std::vector<char> early_seen;      // candidate trees priced while the solves of the others still ran
  std::vector<float> f_pos, f_ang, f_vel, f_pad, f_types;
  std::vector<int16_t> i_typ, i_have;
  std::vector<int32_t> kept, slots, counts;
  mind_aime_plan_out po;
  bool have_plan = false, half_step = false;
  long long plan_gen = -1;          // mind_ctx::pl_gen of the loop's last plan
  std::chrono::steady_clock::time_point t_plan_end;
  // speculative warm start (TrajectoryTreeOptimizer.speculate_warm / _take_speculation, this repo's trajectory_tree.py): the warm-start fits of the
  // PREVIOUS plan's tree shapes run on a second context beside the AIME rounds; a tree whose shape recurs takes its warm-start controls from there
  mind_ctx *side = nullptr;
  hipStream_t side_stream = nullptr;
  struct Shape { std::vector<int32_t> parent; std::vector<float> prob; };
  std::vector<Shape> last_shapes, spec_shapes;
  bool spec_pending = false;
  int spec_skip = 0, spec_backoff = 4, spec_last = -1;
  std::vector<double> spec_xs, spec_us, sol_xs, sol_us, hit_ui, miss_xs, miss_us;
  std::vector<mind_ilqr_stats> spec_st, sol_stw, sol_stf, miss_stw, miss_stf;
  std::vector<float> zero_mean, zero_cov;
  long long warm_speculated = 0, warm_hits = 0;
  double plan_x0[6];
  int last_agents = 0, last_nodes = 0, best = -1, last_agents_plan = 0;
  bool sol_owned = false;           // the last plan's solve results are the loop's own arrays (a speculated cycle), not the context's
  double aime_s = 0, ilqr_s = 0, total_s = 0;
  mind_loop_totals tot;
  bool planned_last = false;
};

namespace {

int loop_row(const mind_loop *L, long long n) {
  if (n < L->d.n_steps) return (int)n;
  return L->d.clamp_last ? L->d.n_steps - 1 : -1;
}

void loop_fill_out(const mind_loop *L, mind_loop_out *o) {
  memset(o, 0, sizeof(*o));
  o->planned = L->planned_last ? 1 : 0;
  o->enabled = L->enabled ? 1 : 0;
  o->n_steps = L->n_steps; o->n_plans = L->n_plans; o->episode_steps = L->ep_steps;
  o->sim_time = L->sim_time; o->last_trigger = L->have_trigger ? L->last_trigger : -1.0;
  memcpy(o->state, L->state, sizeof(o->state));
  memcpy(o->ctrl, L->ctrl, sizeof(o->ctrl));
  if (L->have_plan) {
    o->n_agents = L->last_agents; o->n_trees = L->po.n_trees; o->best = L->best;
    o->n_expanded = L->po.n_expanded; o->n_rounds = L->po.n_rounds;
    o->n_traj_nodes = L->last_nodes;        // (kept by value: the plan's tables are the context's and may be another plan's by now)
    o->costs = L->costs.data();
    o->aime_s = L->aime_s; o->ilqr_s = L->ilqr_s; o->total_s = L->total_s;
  }
  o->tot = L->tot;
}

void loop_sincos(const mind_loop *L, double x, double *sc) {
  if (L->d.sincos_fn) L->d.sincos_fn(x, sc);
  else { sc[0] = sin(x); sc[1] = cos(x); }
}

// MINDPlanner.update_observation (planner.py:50-64) on the tables' row `r`
void loop_observe(mind_loop *L, int r) {
  const int nt = L->d.n_tracks;
  double row[LOOP_ROW];
  // ego: its plant state once enabled, the recording before (ClosedLoopSim._observation); to_object_state (planner.py:61-64).  The recorded
  // ego's entry -- before the take-over and at the take-over step, when the plant state IS the recorded one -- comes from the caller's table
  // (a float32 recording makes numpy evaluate it with its float32 cosine: only the driver's own code reproduces that)
  if (!L->enabled || L->state_recorded) {
    const double *e = L->ego_obs.data() + (size_t)r * 5;
    row[0] = 1.0; row[1] = e[0]; row[2] = e[1]; row[3] = e[2]; row[4] = e[3]; row[5] = e[4];
  } else {
    const double *es = L->state;
    double sc[2];
    loop_sincos(L, es[3], sc);
    row[0] = 1.0; row[1] = es[0]; row[2] = es[1]; row[3] = es[3];
    row[4] = es[2] * sc[1]; row[5] = es[2] * sc[0];
  }
  row[6] = (double)L->timestep[r];
  if (L->obs.empty()) {
    L->obs.emplace_back();
    L->obs[0].track = 0;
    L->slot_of_track[0] = 0;
  }
  const size_t known_before = L->obs.size();
  L->obs[0].push(row);
  // exo agents reported at this step, in the world's order; a track seen for the first time joins the table behind the others
  std::vector<uint8_t> &seen = L->exo_valid;       // (read only)
  const uint8_t *val = seen.data() + (size_t)r * nt;
  for (int i = 1; i < nt; ++i) {
    if (!val[i]) continue;
    int s = L->slot_of_track[i];
    if (s < 0) {
      s = (int)L->obs.size();
      L->obs.emplace_back();
      L->obs[s].track = i;
      L->slot_of_track[i] = s;
    }
    const double *e = L->exo_obs.data() + ((size_t)r * nt + i) * 5;
    row[0] = 1.0; row[1] = e[0]; row[2] = e[1]; row[3] = e[2]; row[4] = e[3]; row[5] = e[4];
    row[6] = (double)L->timestep[r];
    L->obs[s].push(row);
  }
  // tracks not reported: their last state again, unobserved (planner.py:58-62)
  for (size_t s = 1; s < known_before; ++s) {
    LoopTrack &t = L->obs[s];
    if (val[t.track]) continue;
    memcpy(row, t.last(), sizeof(row));
    row[0] = 0.0;
    t.push(row);
  }
}

// kine_propagate (common/kinematics.py:22-36) with the simulator's plant constants (agent.py:298-299)
void loop_propagate(mind_loop *L) {
  const mind_loop_desc &d = L->d;
  const double x = L->state[0], y = L->state[1], v = L->state[2], yaw = L->state[3];
  double a = L->ctrl[0], delta = L->ctrl[1];
  a = fmin(fmax(a, d.max_dec), d.max_acc);
  delta = fmin(fmax(delta, -d.max_steer), d.max_steer);
  const double dt = d.sim_step;
  double o0, o1, o2, o3;
  if (!(L->tan_valid && L->tan_arg == delta)) {        // (the control holds for the steps between two plans)
    L->tan_val = d.tan_fn ? d.tan_fn(delta) : tan(delta);
    L->tan_arg = delta; L->tan_valid = true;
  }
  const double tn = L->tan_val;
  if (L->state_recorded && d.ego_state_is_f32) {
    // the state taken over from a float32 recording is a float32 array: numpy evaluates x + v cos(yaw) dt and y + v sin(yaw) dt in float32 (its
    // float32 cosine / sine of the recorded yaw: the caller's table; the Python float dt is a weak operand), v + a dt in float64 (a is a float64
    // scalar) and yaw + v / wb tan(delta) dt with v / wb in float32 and the rest in float64 (np.tan of a float64 scalar is one); the result
    // array is float64, and so is every later step
    const float *cs = L->ego_trig32.data() + (size_t)L->state_row * 2;
    const float xf = (float)x, yf = (float)y, vf = (float)v, wf = (float)yaw, dtf = (float)dt;
    const float c0 = vf * cs[0], s0 = vf * cs[1];
    const float c1 = c0 * dtf, s1 = s0 * dtf;
    const float q0 = vf / (float)d.wheelbase;
    o0 = (double)(float)(xf + c1); o1 = (double)(float)(yf + s1); o2 = v + a * dt;
    o3 = (double)wf + ((double)q0 * tn) * dt;
  } else {
    double sc[2];
    loop_sincos(L, yaw, sc);
    o0 = x + v * sc[1] * dt; o1 = y + v * sc[0] * dt; o2 = v + a * dt; o3 = yaw + v / d.wheelbase * tn * dt;
  }
  L->state_recorded = false;
  o2 = fmin(fmax(o2, -d.max_speed), d.max_speed);
  L->state[0] = o0; L->state[1] = o1; L->state[2] = o2; L->state[3] = o3;
}

// TrajectoryTreeOptimizer.speculate_warm: the warm-start fit (lane term only: it sees the cost tree's shape, the ego state, the target lane and
// velocity -- no prediction) of every shape of the previous plan, begun on the side context; returns at once
int loop_speculate(mind_loop *L, const double *x0) {
  mind_ctx *c = L->c;
  const mind_loop_desc &d = L->d;
  L->spec_pending = false;
  if (!d.speculative || L->last_shapes.empty()) return MIND_OK;
  if (L->spec_skip > 0) { L->spec_skip -= 1; L->spec_last = -1; return MIND_OK; }
  if (!L->side) {
    if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&L->side_stream, hipStreamNonBlocking) != hipSuccess) return MIND_OK;
    if (mind_ctx_create(c->device, (void *)L->side_stream, &L->side) != MIND_OK) { L->side = nullptr; return MIND_OK; }
  }
  mind_ctx *sc = L->side;
  if (sc->il_finish) (void)mind_ilqr_finish(sc);
  (void)mind_set_tuning(sc, "ilqr_wgs", 1);          // (a fit beside the predictor stays on one workgroup per tree: IlqrCall background)
  L->spec_shapes = L->last_shapes;
  const int n = (int)L->spec_shapes.size();
  size_t M = 0, maxM = 0;
  for (const auto &sh : L->spec_shapes) { M += sh.parent.size(); maxM = std::max(maxM, sh.parent.size()); }
  L->zero_mean.assign(maxM * 2, 0.f); L->zero_cov.assign(maxM, 0.f);        // the warm-start cost has no agent term: one dummy agent per node
  std::vector<mind_cost_tree> trees(n);
  for (int t = 0; t < n; ++t) {
    mind_cost_tree &T = trees[t];
    memset(&T, 0, sizeof(T));
    T.n_nodes = (int)L->spec_shapes[t].parent.size(); T.parent = L->spec_shapes[t].parent.data(); T.prob = L->spec_shapes[t].prob.data();
    T.n_agents = 1; T.agent_mean = L->zero_mean.data(); T.agent_cov = L->zero_cov.data();
  }
  L->spec_xs.resize(M * 6); L->spec_us.resize(M * 2); L->spec_st.resize(n);
  sc->il_begin_only = true;
  const int rc = ilqr_impl(sc, &L->cfg_warm, nullptr, trees.data(), n, x0, L->solve_lane.data(), d.solve_n_lane_pts, d.target_vel, 0, nullptr,
                           L->spec_xs.data(), L->spec_us.data(), L->spec_st.data(), nullptr);
  sc->il_begin_only = false;
  if (rc != MIND_OK) return MIND_OK;                 // (the in-line path then computes -- or reports -- the same thing)
  L->spec_pending = true;
  L->warm_speculated += n;
  L->spec_last = n;
  return MIND_OK;
}

// TrajectoryTreeOptimizer.solve_batch with a speculation in flight: trees whose shape was guessed right run the full fit only (from the
// speculated warm-start controls), the others both fits on the side context beside them.  Results into L->sol_* (tree order).
int loop_solve_speculated(mind_loop *L, const double *x0) {
  mind_ctx *c = L->c, *sc = L->side;
  const mind_loop_desc &d = L->d;
  const int nt = L->po.n_trees, a = L->last_agents_plan;
  const int32_t *off = L->po.tree_off;
  const int M = off[nt];
  // ---- _take_speculation
  bool have = mind_ilqr_finish(sc) == MIND_OK;
  L->spec_pending = false;
  std::vector<int> hit_of(nt, -1);
  std::vector<size_t> spec_off(L->spec_shapes.size() + 1, 0);
  for (size_t j = 0; j < L->spec_shapes.size(); ++j) spec_off[j + 1] = spec_off[j] + L->spec_shapes[j].parent.size();
  int n_hits = 0;
  for (int t = 0; t < nt && have; ++t) {
    const int m = off[t + 1] - off[t];
    for (size_t j = 0; j < L->spec_shapes.size(); ++j) {
      const auto &sh = L->spec_shapes[j];
      if ((int)sh.parent.size() == m && memcmp(sh.parent.data(), L->po.flat_parent + off[t], (size_t)m * 4) == 0 &&
          memcmp(sh.prob.data(), L->po.flat_prob + off[t], (size_t)m * 4) == 0) { hit_of[t] = (int)j; ++n_hits; break; }
    }
  }
  L->warm_hits += n_hits;
  if (L->spec_last >= 0) {
    // a speculated fit only pays when the shape recurs: after a cycle in which fewer than half of the guesses were used, skip the next
    // spec_backoff cycles (doubling after every failed probe, 4 .. 256), then probe again
    if (2 * n_hits < L->spec_last) { L->spec_skip = L->spec_backoff; L->spec_backoff = std::min(2 * L->spec_backoff, 256); }
    else L->spec_backoff = 4;
  }
  L->spec_last = -1;
  L->sol_xs.assign((size_t)M * 6, 0.0); L->sol_us.assign((size_t)M * 2, 0.0); L->sol_stw.assign(nt, mind_ilqr_stats()); L->sol_stf.assign(nt, mind_ilqr_stats());
  auto tree_of = [&](int t) {
    mind_cost_tree T;
    memset(&T, 0, sizeof(T));
    T.n_nodes = off[t + 1] - off[t]; T.parent = L->po.flat_parent + off[t]; T.prob = L->po.flat_prob + off[t];
    T.n_agents = a; T.agent_mean = L->po.flat_mean + (size_t)off[t] * a * 2; T.agent_cov = L->po.flat_cov + (size_t)off[t] * a;
    return T;
  };
  std::vector<mind_cost_tree> hits, misses;
  std::vector<int> hit_idx, miss_idx;
  for (int t = 0; t < nt; ++t) { if (hit_of[t] >= 0) { hit_idx.push_back(t); hits.push_back(tree_of(t)); } else { miss_idx.push_back(t); misses.push_back(tree_of(t)); } }
  int rc = MIND_OK;
  size_t Mm = 0, Mh = 0;
  for (const auto &T : misses) Mm += T.n_nodes;
  for (const auto &T : hits) Mh += T.n_nodes;
  bool miss_on_side = false;
  if (!misses.empty()) {          // both fits, as without speculation: beside the hits' full fits when there are any
    L->miss_xs.resize(Mm * 6); L->miss_us.resize(Mm * 2); L->miss_stw.resize(misses.size()); L->miss_stf.resize(misses.size());
    if (!hits.empty()) {
      (void)mind_set_tuning(sc, "ilqr_wgs", c->ilqr_wgs);
      rc = mind_ilqr_contingency_begin(sc, &L->cfg_warm, &L->cfg_full, misses.data(), (int)misses.size(), x0, L->solve_lane.data(), d.solve_n_lane_pts, d.target_vel,
                                       L->miss_xs.data(), L->miss_us.data(), L->miss_stw.data(), L->miss_stf.data());
      if (rc) return fail(c, rc, "mind_loop: %s", sc->err.c_str());
      miss_on_side = true;
    } else {
      if ((rc = mind_ilqr_contingency(c, &L->cfg_warm, &L->cfg_full, misses.data(), (int)misses.size(), x0, L->solve_lane.data(), d.solve_n_lane_pts, d.target_vel,
                                      L->miss_xs.data(), L->miss_us.data(), L->miss_stw.data(), L->miss_stf.data())))
        return rc;
    }
  }
  if (!hits.empty()) {            // warm-start controls are there already: full fit only
    L->hit_ui.resize(Mh * 2);
    size_t o = 0;
    for (int t : hit_idx) {
      const int m = off[t + 1] - off[t];
      memcpy(L->hit_ui.data() + o * 2, L->spec_us.data() + spec_off[hit_of[t]] * 2, (size_t)m * 2 * sizeof(double));
      o += m;
    }
    std::vector<double> hx(Mh * 6), hu(Mh * 2);
    std::vector<mind_ilqr_stats> hs(hits.size());
    if ((rc = mind_ilqr_solve_trees(c, &L->cfg_full, hits.data(), (int)hits.size(), x0, L->solve_lane.data(), d.solve_n_lane_pts, d.target_vel, 1, L->hit_ui.data(),
                                    hx.data(), hu.data(), hs.data()))) {
      if (miss_on_side) (void)mind_ilqr_finish(sc);
      return rc;
    }
    o = 0;
    for (size_t k = 0; k < hit_idx.size(); ++k) {
      const int t = hit_idx[k], m = off[t + 1] - off[t];
      memcpy(L->sol_xs.data() + (size_t)off[t] * 6, hx.data() + o * 6, (size_t)m * 6 * sizeof(double));
      memcpy(L->sol_us.data() + (size_t)off[t] * 2, hu.data() + o * 2, (size_t)m * 2 * sizeof(double));
      L->sol_stw[t] = L->spec_st[hit_of[t]]; L->sol_stf[t] = hs[k];
      o += m;
    }
  }
  if (miss_on_side && (rc = mind_ilqr_finish(sc))) return fail(c, rc, "mind_loop: %s", sc->err.c_str());
  size_t o = 0;
  for (size_t k = 0; k < miss_idx.size(); ++k) {
    const int t = miss_idx[k], m = off[t + 1] - off[t];
    memcpy(L->sol_xs.data() + (size_t)off[t] * 6, L->miss_xs.data() + o * 6, (size_t)m * 6 * sizeof(double));
    memcpy(L->sol_us.data() + (size_t)off[t] * 2, L->miss_us.data() + o * 2, (size_t)m * 2 * sizeof(double));
    L->sol_stw[t] = L->miss_stw[k]; L->sol_stf[t] = L->miss_stf[k];
    o += m;
  }
  return MIND_OK;
}

// MINDPlanner.plan (planner.py:66-145) of the enabled agent
int loop_plan(mind_loop *L) {
  mind_ctx *c = L->c;
  const mind_loop_desc &d = L->d;
  const auto t0 = std::chrono::steady_clock::now();
  // MIND_LOOP_TRACE=1: host time stamps of the cycle's sections on stderr (diagnostic, as MIND_PLAN_TRACE inside the plan)
  static const bool loop_trace = getenv("MIND_LOOP_TRACE") != nullptr;
  auto TRL = [&](const char *what) {
    if (loop_trace) fprintf(stderr, "[loop] %8.1f us  %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), what);
  };
  if (loop_trace && L->have_plan) fprintf(stderr, "[loop] %8.1f us since the last plan's end (evaluation excluded: steps, observation)\n",
                                          std::chrono::duration<double, std::micro>(t0 - L->t_plan_end).count());
  // ---- get_agent_trajectories (utils.py:245-342): AV first, tracks whose last state is unobserved are dropped, windows left-padded
  L->kept.clear(); L->slots.clear();
  for (size_t s = 0; s < L->obs.size(); ++s) {
    const LoopTrack &t = L->obs[s];
    if (t.last()[0] == 0.0) continue;
    L->kept.push_back(t.track);
    L->slots.push_back(L->type_slot[t.track]);
  }
  const int a = (int)L->kept.size(), T = LOOP_OBS;
  L->raw.assign((size_t)a * T * 6, 0.0);
  for (int i = 0; i < a; ++i) {
    const LoopTrack &t = L->obs[L->slot_of_track[L->kept[i]]];
    for (int k = 0; k < t.count; ++k)
      memcpy(L->raw.data() + ((size_t)i * T + (T - t.count + k)) * 6, t.rows[(t.head + k) % LOOP_OBS], 6 * sizeof(double));
  }
  L->f_pos.resize((size_t)a * T * 2); L->f_ang.resize((size_t)a * T); L->f_vel.resize((size_t)a * T * 2);
  L->i_typ.resize((size_t)a * T * 7); L->i_have.resize((size_t)a * T);
  int rc = mind_fill_tracks(L->raw.data(), a, T, L->slots.data(), L->f_pos.data(), L->f_ang.data(), L->f_vel.data(), L->i_typ.data(), L->i_have.data());
  if (rc) return fail(c, rc, "mind_loop: mind_fill_tracks failed");
  TRL("tracks filled");
  L->f_pad.resize((size_t)a * T); L->f_types.resize((size_t)a * T * 7);
  for (size_t i = 0; i < L->f_pad.size(); ++i) L->f_pad[i] = (float)L->i_have[i];
  for (size_t i = 0; i < L->f_types.size(); ++i) L->f_types[i] = (float)L->i_typ[i];
  // ---- mind_aime_plan, device-built root, the contingency solves begun behind it (scenario_tree.py:38-58, planner.py:174-178)
  mind_aime_plan_in pi;
  memset(&pi, 0, sizeof(pi));
  pi.n_agents = a; pi.n_lanes = d.n_lanes; pi.n_lane_pts = d.n_lane_pts;
  pi.raw_pos = L->f_pos.data(); pi.raw_ang = L->f_ang.data(); pi.raw_vel = L->f_vel.data(); pi.raw_pad = L->f_pad.data();
  pi.types = L->f_types.data();
  pi.target_lane = L->target_lane.data(); pi.target_lane_info = L->target_lane_info.data();
  pi.lane_pts = L->lane_pts.data(); pi.lane_flags = L->lane_flags.data();
  const double cur_vel = L->state[2];
  pi.travel0 = (float)((cur_vel >= (double)d.min_vel ? cur_vel : (double)d.min_vel) * d.time_ahead);      // F32(max(cur_vel, 0.5) * tar_time_ahead)
  pi.time_ahead = (float)d.time_ahead; pi.min_vel = d.min_vel; pi.dist_thres = d.dist_thres;
  pi.max_depth = d.max_depth; pi.max_rounds = d.max_rounds; pi.pred_len = d.pred_len; pi.prob_floor = d.prob_floor;
  double *x0 = L->plan_x0;
  x0[0] = L->state[0]; x0[1] = L->state[1]; x0[2] = L->state[2]; x0[3] = L->state[3]; x0[4] = L->ctrl[0]; x0[5] = L->ctrl[1];
  // warm-start fits of the previous plan's tree shapes start now, beside the predictor; with one in flight the plan does not begin the
  // solves itself (TrajectoryTreeOptimizer.plan_solve_args returns None then)
  if ((rc = loop_speculate(L, x0))) return rc;
  if (!L->spec_pending) {
    pi.solve_cfg_warm = &L->cfg_warm; pi.solve_cfg_full = &L->cfg_full;
    pi.solve_x0 = x0; pi.solve_lane = L->solve_lane.data(); pi.solve_n_lane_pts = d.solve_n_lane_pts; pi.solve_target_vel = d.target_vel;
  }
  L->have_plan = false;
  TRL("plan call begins");
  rc = mind_aime_plan(c, &pi, &L->po);
  if (rc) return rc;
  const auto t1 = std::chrono::steady_clock::now();
  TRL("plan returned (solves begun)");
  const int nt = L->po.n_trees;
  if (nt <= 0) return fail(c, MIND_ESTATE, "mind_loop: the plan returned no scenario tree");
  const int32_t *off = L->po.tree_off;
  const int M = off[nt];
  bool early_done = false;        // the candidates were priced while the solves ran (below)
  const double *xs, *us;
  const mind_ilqr_stats *stw, *stf;
  L->last_agents_plan = a;
  if (L->spec_pending) {
    if ((rc = loop_solve_speculated(L, x0))) return rc;
    xs = L->sol_xs.data(); us = L->sol_us.data(); stw = L->sol_stw.data(); stf = L->sol_stf.data();
    L->sol_owned = true;
  } else {
    if (!L->po.solves_begun) return fail(c, MIND_ESTATE, "mind_loop: the plan could not begin its contingency solves (%s)", c->err.c_str());
    // ---- collect the solves (mind_ilqr_finish_plan without the copy: the results stay in the context until its next plan)
    if (!c->il_finish || !c->il_finish_owned) return fail(c, MIND_ESTATE, "mind_loop: no plan-begun tree-iLQR call is pending");
    // ---- a launch of small trees writes every tree's results to the host and marks it when they are complete: price the candidates as they
    //      arrive (evaluate_traj_tree of one tree is independent of the others), beside the trees the device is still solving -- only the
    //      last one's evaluation stays between the kernel's end and the chosen control
    const mind_ctx::IlEarly E = c->il_early;
    if (c->early_eval && E.done && E.n_trees == nt && E.nodes == (long)M) {
      L->costs.assign(nt, 0.0);
      L->early_seen.assign(nt, 0);
      const void *elane_e = d.eval_lane_is_f32 ? (const void *)L->eval_lane_f32.data() : (const void *)L->eval_lane_f64.data();
      int left = nt;
      const auto te0 = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (left > 0) {
        bool any = false;
        for (int t = 0; t < nt; ++t) {
          if (L->early_seen[t] || __atomic_load_n((const unsigned *)&E.done[t], __ATOMIC_ACQUIRE) != E.gen) continue;
          const int m = off[t + 1] - off[t];
          L->eval_st.resize((size_t)(m + 1) * 6); L->eval_ct.resize((size_t)(m + 1) * 2);
          memcpy(L->eval_st.data(), x0, 6 * sizeof(double));
          L->eval_ct[0] = 0.0; L->eval_ct[1] = 0.0;
          memcpy(L->eval_st.data() + 6, E.xs + (size_t)off[t] * 6, (size_t)m * 6 * sizeof(double));
          memcpy(L->eval_ct.data() + 2, E.us + (size_t)off[t] * 2, (size_t)m * 2 * sizeof(double));
          const int32_t cnt1 = m + 1;
          if ((rc = mind_eval_traj_trees(L->eval_st.data(), L->eval_ct.data(), &cnt1, 1, elane_e, d.eval_lane_is_f32, d.eval_n_lane_pts, d.target_vel, &L->costs[t])))
            return fail(c, rc, "mind_loop: mind_eval_traj_trees failed");
          L->early_seen[t] = 1; --left; any = true;
        }
        if (!any && (++spins & 0x3ffu) == 0u &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - te0).count() > 0.25) break;      // (never hang on a word: the usual path below)
      }
      early_done = left == 0;
    }
    if ((rc = mind_ilqr_finish(c))) return rc;
    if ((size_t)M * 6 != c->pl_sol_xs.size() || (size_t)nt != c->pl_sol_stf.size()) return fail(c, MIND_ESTATE, "mind_loop: the solves' results do not fit the plan");
    xs = c->pl_sol_xs.data(); us = c->pl_sol_us.data(); stw = c->pl_sol_stw.data(); stf = c->pl_sol_stf.data();
    L->sol_owned = false;
  }
  const auto t2 = std::chrono::steady_clock::now();
  TRL("solves collected");
  if (d.speculative) {           // the shapes the next cycle speculates on (_last_structs)
    L->last_shapes.resize(nt);
    for (int t = 0; t < nt; ++t) {
      L->last_shapes[t].parent.assign(L->po.flat_parent + off[t], L->po.flat_parent + off[t + 1]);
      L->last_shapes[t].prob.assign(L->po.flat_prob + off[t], L->po.flat_prob + off[t + 1]);
    }
  }
  // ---- evaluate_traj_tree of every candidate (planner.py:180-198): the tree's nodes = the root (x0, zero control) + its trajectory nodes
  L->counts.resize(nt);
  if (early_done) { for (int t = 0; t < nt; ++t) L->counts[t] = off[t + 1] - off[t] + 1; }
  else {
  L->eval_st.resize((size_t)(M + nt) * 6); L->eval_ct.resize((size_t)(M + nt) * 2); L->costs.resize(nt);
  size_t o = 0;
  for (int t = 0; t < nt; ++t) {
    const int m = off[t + 1] - off[t];
    memcpy(L->eval_st.data() + o * 6, x0, 6 * sizeof(double));
    L->eval_ct[o * 2] = 0.0; L->eval_ct[o * 2 + 1] = 0.0;
    memcpy(L->eval_st.data() + (o + 1) * 6, xs + (size_t)off[t] * 6, (size_t)m * 6 * sizeof(double));
    memcpy(L->eval_ct.data() + (o + 1) * 2, us + (size_t)off[t] * 2, (size_t)m * 2 * sizeof(double));
    L->counts[t] = m + 1;
    o += (size_t)m + 1;
  }
  const void *elane = d.eval_lane_is_f32 ? (const void *)L->eval_lane_f32.data() : (const void *)L->eval_lane_f64.data();
  if ((rc = mind_eval_traj_trees(L->eval_st.data(), L->eval_ct.data(), L->counts.data(), nt, elane, d.eval_lane_is_f32, d.eval_n_lane_pts, d.target_vel, L->costs.data())))
    return fail(c, rc, "mind_loop: mind_eval_traj_trees failed");
  }
  // ---- the reference's strict `<` scan (planner.py:131-136): the first minimum, a NaN never wins
  int best = -1;
  double min_cost = INFINITY;
  for (int t = 0; t < nt; ++t)
    if (L->costs[t] < min_cost) { min_cost = L->costs[t]; best = t; }
  if (best < 0) return fail(c, MIND_ESTATE, "mind_loop: no candidate tree has a finite cost");
  // first control: (a, delta) of the state of the root's first child (planner.py:138-141, Q15)
  int first = -1;
  for (int k = off[best]; k < off[best + 1]; ++k)
    if (L->po.flat_parent[k] == -1) { first = k; break; }
  if (first < 0) return fail(c, MIND_ESTATE, "mind_loop: the chosen tree has no root child");
  L->ctrl[0] = xs[(size_t)first * 6 + 4]; L->ctrl[1] = xs[(size_t)first * 6 + 5];
  L->best = best; L->last_agents = a; L->last_nodes = M; L->have_plan = true; L->plan_gen = c->pl_gen;
  // accounting (TrajectoryTreeOptimizer.counters, MINDPlanner.timing_sum; bench.py's live kernel durations when profiling is on)
  mind_loop_totals &S = L->tot;
  for (int t = 0; t < nt; ++t) {
    const long long m = off[t + 1] - off[t], iw = stw[t].iterations, jf = stf[t].iterations;
    S.iterations += iw + jf; S.node_iterations += m * (iw + jf); S.node_iterations_exo += m * jf * a;
  }
  S.warm_speculated = L->warm_speculated; S.warm_hits = L->warm_hits;
  S.plans += 1; S.expansions += L->po.n_expanded; S.scen_trees += nt; S.rounds += L->po.n_rounds;
  const double N = (double)(a + d.n_lanes + 1);
  S.scene_n2 += (double)L->po.n_expanded * N * N; S.scene_n_a1 += (double)L->po.n_expanded * N * (double)(a + 1);
  if (c->profiling) {
    S.pair_ms += L->po.pair_ms; S.pair_launches += L->po.pair_launches;
    if (c->ilqr_ms > 0.f) {
      S.ilqr_ms += c->ilqr_ms; S.ilqr_launches += 1; S.ilqr_trees += c->ilqr_trees; S.ilqr_workgroups_per_tree = c->ilqr_multi;
      for (int q = 0; q < 9; ++q) S.ilqr_prof[q] += c->il_prof[q];
      S.ilqr_node_steps += c->il_prof[2] * c->il_prof[1];
    }
  }
  const auto t3 = std::chrono::steady_clock::now();
  TRL("evaluated, control chosen");
  L->t_plan_end = t3;
  L->aime_s = std::chrono::duration<double>(t1 - t0).count();
  L->ilqr_s = std::chrono::duration<double>(t2 - t1).count();
  L->total_s = std::chrono::duration<double>(t3 - t0).count();
  S.aime_s += L->aime_s; S.ilqr_s += L->ilqr_s; S.total_s += L->total_s;
  return MIND_OK;
}

// ClosedLoopSim.step: returns MIND_OK, *planned set
int loop_step(mind_loop *L) {
  mind_ctx *c = L->c;
  const mind_loop_desc &d = L->d;
  L->planned_last = false;
  const int r = loop_row(L, L->ep_steps);
  if (r < 0) return fail(c, MIND_EINVAL, "mind_loop: the scene tables end at step %d (episode step %lld)", d.n_steps, L->ep_steps);
  if (L->half_step) return fail(c, MIND_ESTATE, "mind_loop: the plan of the last step failed after its observation update: take the loop over (mind_loop_export) or reset it");
  {
    if (L->sim_time >= d.enable_time && !L->enabled) {            // check_enable: take over from the recording
      L->enabled = true;
      memcpy(L->state, L->ego_state.data() + (size_t)r * 4, 4 * sizeof(double));
      L->state_recorded = true; L->state_row = r;
      L->ctrl[0] = 0.0; L->ctrl[1] = 0.0;
    }
    if (!L->have_trigger || (L->sim_time - L->last_trigger) >= d.plan_step) {
      L->have_trigger = true; L->last_trigger = L->sim_time;
      loop_observe(L, r);
      if (L->enabled) {
        const int rc = loop_plan(L);
        if (rc) {
          // the observation update of this step is done and must not be repeated: a caller that takes the loop over finishes the step itself
          L->half_step = true;
          return rc;
        }
        L->n_plans += 1;
        L->planned_last = true;
      }
    }
  }
  if (L->enabled) loop_propagate(L);
  L->sim_time += d.sim_step;
  L->n_steps += 1; L->ep_steps += 1;
  return MIND_OK;
}

}  // namespace

extern "C" int mind_loop_reset(mind_loop *L);
extern "C" int mind_loop_create(mind_ctx *c, const mind_loop_desc *d, mind_loop **out) {
  if (!c || !d || !out) return MIND_EINVAL;
  if (!c->have_weights) return fail(c, MIND_ESTATE, "weights not loaded");
  if (d->n_tracks <= 0 || d->n_steps <= 0 || !d->ego_state || !d->ego_obs || (d->ego_state_is_f32 && !d->ego_trig32) || !d->exo_obs || !d->exo_valid || !d->timestep || !d->type_slot || !(d->sim_step > 0.0) ||
      !(d->plan_step > 0.0) || !(d->wheelbase > 0.0) || d->n_lanes <= 0 || !d->lane_pts || !d->lane_flags || d->n_lane_pts < 12 || !d->target_lane ||
      !d->target_lane_info || !d->cfg_warm || !d->cfg_full || d->solve_n_lane_pts < 2 || !d->solve_lane || d->eval_n_lane_pts < 2 || !d->eval_lane ||
      d->max_rounds <= 0 || d->max_rounds > 32 || d->pred_len < 2 || d->pred_len > 60)
    return fail(c, MIND_EINVAL, "mind_loop_create: bad argument");
  mind_loop *L = new (std::nothrow) mind_loop();
  if (!L) return fail(c, MIND_ENOMEM, "mind_loop_create: out of memory");
  L->c = c; L->d = *d;
  const size_t S = (size_t)d->n_steps, nt = (size_t)d->n_tracks;
  L->ego_state.assign(d->ego_state, d->ego_state + S * 4);
  L->ego_obs.assign(d->ego_obs, d->ego_obs + S * 5);
  if (d->ego_trig32) L->ego_trig32.assign(d->ego_trig32, d->ego_trig32 + S * 2);
  L->exo_obs.assign(d->exo_obs, d->exo_obs + S * nt * 5);
  L->exo_valid.assign(d->exo_valid, d->exo_valid + S * nt);
  L->timestep.assign(d->timestep, d->timestep + S);
  L->type_slot.assign(d->type_slot, d->type_slot + nt);
  L->lane_pts.assign(d->lane_pts, d->lane_pts + (size_t)d->n_lanes * 22);
  L->lane_flags.assign(d->lane_flags, d->lane_flags + (size_t)d->n_lanes * 6);
  L->target_lane.assign(d->target_lane, d->target_lane + (size_t)d->n_lane_pts * 2);
  L->target_lane_info.assign(d->target_lane_info, d->target_lane_info + (size_t)d->n_lane_pts * 12);
  L->solve_lane.assign(d->solve_lane, d->solve_lane + (size_t)d->solve_n_lane_pts * 2);
  if (d->eval_lane_is_f32) L->eval_lane_f32.assign((const float *)d->eval_lane, (const float *)d->eval_lane + (size_t)d->eval_n_lane_pts * 2);
  else L->eval_lane_f64.assign((const double *)d->eval_lane, (const double *)d->eval_lane + (size_t)d->eval_n_lane_pts * 2);
  L->cfg_warm = *d->cfg_warm; L->cfg_full = *d->cfg_full;
  // (the descriptor's pointers are not used after this call)
  L->d.ego_state = nullptr; L->d.ego_obs = nullptr; L->d.ego_trig32 = nullptr; L->d.exo_obs = nullptr; L->d.exo_valid = nullptr; L->d.timestep = nullptr; L->d.type_slot = nullptr; L->d.lane_pts = nullptr;
  L->d.lane_flags = nullptr; L->d.target_lane = nullptr; L->d.target_lane_info = nullptr; L->d.cfg_warm = nullptr; L->d.cfg_full = nullptr;
  L->d.solve_lane = nullptr; L->d.eval_lane = nullptr;
  L->slot_of_track.assign(nt, -1);
  memset(&L->po, 0, sizeof(L->po));
  memset(&L->tot, 0, sizeof(L->tot));
  (void)mind_loop_reset(L);
  *out = L;
  return MIND_OK;
}

extern "C" int mind_loop_destroy(mind_loop *L) {
  if (L && L->side) {
    (void)mind_ctx_destroy(L->side);
    if (L->side_stream) (void)hipStreamDestroy(L->side_stream);
  }
  delete L;
  return MIND_OK;
}

extern "C" int mind_loop_reset(mind_loop *L) {
  if (!L) return MIND_EINVAL;
  L->sim_time = 0.0; L->have_trigger = false; L->last_trigger = -1.0; L->enabled = false; L->half_step = false;
  memcpy(L->state, L->ego_state.data(), 4 * sizeof(double));        // world.agent_state(0, 0.0)
  L->state_recorded = false; L->tan_valid = false;
  if (L->spec_pending && L->side) { (void)mind_ilqr_finish(L->side); L->spec_pending = false; }
  L->ctrl[0] = 0.0; L->ctrl[1] = 0.0;
  L->ep_steps = 0;
  L->obs.clear();
  std::fill(L->slot_of_track.begin(), L->slot_of_track.end(), -1);
  L->planned_last = false;
  return MIND_OK;
}

extern "C" int mind_loop_advance(mind_loop *L, int until_plans, double until_time, long long max_steps, mind_loop_out *out) {
  if (!L || !out || max_steps < 0) return MIND_EINVAL;
  if (L->c->xfn) return fail(L->c, MIND_ESTATE, "mind_loop_advance: the context is sharded (mind_set_exchange): the native loop plans on one GPU");
  int done = 0, rc = MIND_OK;
  bool planned_any = false;
  for (long long s = 0; s < max_steps; ++s) {
    if (until_time >= 0.0 && !(L->sim_time < until_time - 1e-9)) break;
    if ((rc = loop_step(L))) break;
    if (L->planned_last) {
      planned_any = true;
      if (until_plans > 0 && ++done >= until_plans) break;
    }
  }
  L->planned_last = planned_any;
  loop_fill_out(L, out);
  return rc;
}

extern "C" int mind_loop_state(mind_loop *L, mind_loop_out *out) {
  if (!L || !out) return MIND_EINVAL;
  loop_fill_out(L, out);
  return MIND_OK;
}

extern "C" int mind_loop_last_plan(mind_loop *L, mind_aime_plan_out *plan, const double **xs, const double **us, const mind_ilqr_stats **stats_warm,
                                   const mind_ilqr_stats **stats_full, const int32_t **agent_tracks, const float **types, double *x0) {
  if (!L || !plan) return MIND_EINVAL;
  if (!L->have_plan) return fail(L->c, MIND_ESTATE, "mind_loop_last_plan: the loop holds no plan");
  if (L->plan_gen != L->c->pl_gen) return fail(L->c, MIND_ESTATE, "mind_loop_last_plan: the context has planned again since (another planner shares it): the loop's plan tables are gone");
  *plan = L->po;
  if (xs) *xs = L->sol_owned ? L->sol_xs.data() : L->c->pl_sol_xs.data();
  if (us) *us = L->sol_owned ? L->sol_us.data() : L->c->pl_sol_us.data();
  if (stats_warm) *stats_warm = L->sol_owned ? L->sol_stw.data() : L->c->pl_sol_stw.data();
  if (stats_full) *stats_full = L->sol_owned ? L->sol_stf.data() : L->c->pl_sol_stf.data();
  if (agent_tracks) *agent_tracks = L->kept.data();
  if (types) *types = L->f_types.data();
  if (x0) memcpy(x0, L->plan_x0, 6 * sizeof(double));
  return MIND_OK;
}

extern "C" int mind_loop_export(mind_loop *L, int cap, int *n, int32_t *track, int32_t *count, double *rows) {
  if (!L || !n) return MIND_EINVAL;
  *n = (int)L->obs.size();
  if (*n > cap || !track || !count || !rows) return MIND_EINVAL;
  for (int s = 0; s < *n; ++s) {
    const LoopTrack &t = L->obs[s];
    track[s] = t.track; count[s] = t.count;
    for (int k = 0; k < t.count; ++k)
      memcpy(rows + ((size_t)s * LOOP_OBS + k) * LOOP_ROW, t.rows[(t.head + k) % LOOP_OBS], LOOP_ROW * sizeof(double));
  }
  return MIND_OK;
}
