"""GPU: one full MINDPlanner.plan() (HIP predictor over the AIME tree + HIP tree-iLQR) on synthetic
worlds against the reference's plan() captured in tests/golden/plan.npz (formula weights)."""
import os
import sys

import numpy as np
import pytest

from mind_amd.synth import SynthWorld

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = dict(np.load(os.path.join(ROOT, "tests", "golden", "plan.npz")))
CFG = os.path.join(ROOT, "mind_amd", "planners", "mind", "configs", "synthetic.json")


def make_planner(world):
    from mind_amd.planners.mind.planner import MINDPlanner
    pl = MINDPlanner(CFG)
    for s in range(50):
        pl.update_observation(world.local_semantic_map(round(0.1 * s, 6)))
    lcl = world.local_semantic_map(4.9)
    pl.update_target_lane(np.asarray(world.target_lane[::2], dtype=np.float64))
    pl.update_state_ctrl(lcl.ego_agent.state, np.array([0.0, 0.0]))
    return pl, lcl


@pytest.mark.parametrize("name,wkw", [("p6", dict(n_agents=6, n_lanes=3, n_segs=8, seed=1)),
                                      ("p12", dict(n_agents=12, n_lanes=3, n_segs=10, seed=2))])
def test_plan_matches_reference(name, wkw, hip_predictor):
    pl, lcl = make_planner(SynthWorld(**wkw))
    ok, ctrl, (st, tt) = pl.plan(lcl)
    assert ok
    st, tt = st[0], tt[0]
    keys = list(st.nodes.keys())
    assert keys == list(G[name + "_scen_keys"])                       # same surviving branch (selection index)
    probs = np.array([float(np.ravel(st.nodes[k].data[0])[0]) for k in keys])
    assert np.abs(probs - G[name + "_scen_probs"]).max() < 1e-5
    for k in keys:
        assert np.abs(st.nodes[k].data[1][:, ::5] - G[f"{name}_scen_{k}_pos"]).max() < 1e-3     # agent trajectories [m]
        assert np.abs(st.nodes[k].data[2][:, ::5] - G[f"{name}_scen_{k}_cov"]).max() < 1e-3
    tk = [k for k in tt.nodes.keys() if k != -1]
    assert np.array_equal(np.array([tt.nodes[k].parent_key for k in tk]), G[name + "_traj_parent"])
    xs = np.array([tt.nodes[k].data[0] for k in tk])
    assert np.abs(xs - G[name + "_traj_xs"]).max() < 1e-3                                        # ego trajectory [m]
    assert np.abs(np.asarray(ctrl) - G[name + "_ctrl"]).max() < 1e-3


def test_plan_repeatable_and_timed(hip_predictor):
    pl, lcl = make_planner(SynthWorld(n_agents=6, n_lanes=3, n_segs=8, seed=1))
    r1 = pl.plan(lcl)
    r2 = pl.plan(lcl)
    assert np.array_equal(r1[1], r2[1])
    assert pl.timing["nodes_expanded"] >= 1 and pl.timing["total_s"] < 5.0


def test_concurrent_scenes_equal_sequential_runs():
    """BASELINE config 3: scenes planned concurrently (one host thread + HIP context + stream per scene) give exactly
    the controls they give when planned alone."""
    import sys
    import threading
    import torch
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop

    def run(seed_off, out, key, in_stream):
        def body():
            wkw = dict(WORKLOADS["demo1"])
            wkw["seed"] += seed_off
            pl, sim, w = make_closed_loop(wkw)
            ctrls = []
            for _ in range(3):
                sim.run_plans(1)
                ctrls.append(np.array(pl.ctrl, dtype=np.float64))
            out[key] = np.stack(ctrls)
        if in_stream:
            with torch.cuda.stream(torch.cuda.Stream()):
                body()
                torch.cuda.current_stream().synchronize()
        else:
            body()

    alone, conc = {}, {}
    for i in range(2):
        run(i, alone, i, False)
    ths = [threading.Thread(target=run, args=(i, conc, i, True)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i in range(2):
        assert np.array_equal(alone[i], conc[i]), i
    assert not np.array_equal(alone[0], alone[1])


def test_bench_prints_one_contract_json_line(tmp_path):
    """bench.py contract: exactly one COMPACT JSON line on stdout (< 4 KB: the driver could not take round 4's 20 KB line) with the
    driver's keys, the roofline and cpu_baseline objects and the few scalars of the other measurements; every block in full in the extras file
    the line names."""
    import json
    import subprocess
    import sys
    extras = str(tmp_path / "bench_extras.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--tree-steps", "2"],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, MIND_BENCH_EXTRAS=extras))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    assert len(lines[0]) < 4096, len(lines[0])
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "sim steps/s" and d["value"] > 83.0 and "workload" in d["config"]
    assert "recorded scene demo_1" in d["config"]["workload"] and d["data"].startswith("recorded AV2 scene")     # BASELINE configs[1]
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 * r["frac"] and r["unit"] in ("TFLOP/s", "GB/s")
    assert 0 < r["frac"] <= 1.0 and 0 < r["mfma_frac"] <= 1.0 and 0 < r["hbm_frac"] <= 1.0 and r["avg_launch_ms"] > 0
    # traffic: HBM bytes per launch from two sibling rocprofv3 --pmc passes (null only where rocprofv3 is missing)
    assert r["traffic"] is None or r["traffic"] > 0.5 * r["algorithmic_bytes_per_launch"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["value_1_thread"] > 0
    assert 0 < d["k_ilqr"]["share_of_step"] < 1 and d["k_ilqr"]["kernel_ms_per_launch"] > 0
    # the credited slots carry an fp32-class arithmetic (the reference computes in fp32); the two-way split rides beside it
    assert d["dtype"].startswith(("f32", "f32-class")) and r["arith"] in ("f32", "bf16x6")
    assert d["bf16x3"]["value"] > 83.0 and d["tree"]["ms_per_plan"] > 0 and 0 < d["tree"]["k_pair"]["hbm_frac"] <= 1.0
    assert d["tree_f32"]["ms_per_plan"] > 0 and 0 < d["tree_f32"]["k_pair"]["mfma_frac"] <= 1.0
    assert d["config"]["weights"] == "formula_branching:20240121" and d["config"]["expansions_per_plan"] >= 2      # a real AIME tree
    # the extras file: every block of the run in full
    x = json.load(open(extras))
    assert d["extras_file"] == extras and abs(x["value"] - d["value"]) < 1e-3 * d["value"]
    k = x["k_ilqr"]
    assert k["cycles_per_node_step"]["riccati"] > 100 and abs(sum(k["phase_share"].values()) - 1) < 1e-6
    assert x["roofline"]["traffic"] is None or x["roofline"]["traffic_detail"]["launches_counted"] > 0
    assert x["stress"]["expansions_per_plan"] == 259 and x["stress_bf16"]["expansions_per_plan"] == 259
    assert x["stress_deep"]["expansions_per_plan"] == 1555 and x["stress_deeper"]["expansions_per_plan"] == 9331
    assert x["plain_formula_weights"]["expansions_per_plan"] == 1.0
    # BASELINE configs[2] in the default line: four recorded scenes concurrently, a process per scene and one thread's event loop
    assert d["config3"]["scenes"] == 4 and d["config3"]["sim_steps_per_s"] > 83.0 and d["config3"]["processes"] > 0 and d["config3"]["one_thread_event_loop"] > 0 and d["config3"]["x16_two_processes"] > 0
    assert x["synthetic_branching"]["expansions_per_plan"] >= 2 and x["tree"]["expansions_per_plan"] == 259
    assert x["tree"]["k_pair"]["frac"] <= 1.0 and set(x["recorded_scenes"]) >= {"demo_2", "demo_3", "demo_4", "demo_1_whole_run"}


class _arithmetic:
    """run a block with the predictor's contractions in the named arithmetic ("bf16x6": the default, the reference's fp32 class on the bf16 MFMA;
    "f32": plain fp32 operands on the fp32 MFMA; "bf16x3": the opt-in two-way split) and put the thread's runtime back afterwards"""

    def __init__(self, rt, prec):
        self.rt, self.prec = rt, prec

    def __enter__(self):
        self.before = self.rt.pair_precision()
        self.rt.set_pair_precision(self.prec)
        assert self.rt.pair_precision() == self.prec

    def __exit__(self, *exc):
        self.rt.set_pair_precision(self.before)


_WAIVED_PATH = os.path.join(ROOT, "tests", "golden", "waived_cycles.json")


def _check_waived(test, key, observed):
    """The cycles a parity test does NOT compare outright are classified (better optimum under the same objective / a fit whose own
    solution moves by more than the tolerance under its inputs' rounding noise) -- and pinned: the classified sets must equal the record
    committed in tests/golden/waived_cycles.json, so that no cycle can silently join or leave them.  A change that moves last bits moves
    the chaotic fits: re-run with MIND_UPDATE_WAIVED=<file> (the observed sets are written there), review the difference, commit the file."""
    import json
    observed = {k: sorted(int(x) for x in v) for k, v in observed.items()}
    upd = os.environ.get("MIND_UPDATE_WAIVED")
    if upd:
        rec = json.load(open(upd)) if os.path.exists(upd) else {}
        rec.setdefault(test, {})[key] = observed
        with open(upd, "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True)
        return
    rec = json.load(open(_WAIVED_PATH))          # ("_meta": the toolchain the records were pinned on and the regeneration command)
    want = rec.get(test, {}).get(key)
    if os.environ.get("MIND_PRINT_WAIVED") and observed != want:
        # a legitimate last-bit change (compiler, driver, tuning) moves the chaotic fits: show every set that moved instead of stopping at the first
        print(f"[waived] {test} / {key}: observed {observed} | pinned {want}")
        return
    assert want is not None, f"no pinned record for {test} / {key}: observed {observed}"
    assert observed == want, (f"{test} / {key}: classified cycles {observed} != the pinned record {want} (tests/golden/waived_cycles.json; its _meta entry "
                              f"names the toolchain the record was made on and how to regenerate it for review)")


@pytest.mark.parametrize("prec", ["bf16x3", "f32", "bf16x6"])
@pytest.mark.parametrize("scene", ["demo_1", "demo_2", "demo_3", "demo_4"])
def test_recorded_demo_scenes_match_reference_closed_loop(scene, prec):
    """North-star parity on the reference's four recorded AV2 scenes: the reference's own simulator loop (headless,
    CPU, formula weights -- its trained checkpoint is not in the tree) was run to the first four planning cycles
    (tests/golden/gen_golden.py demo_plans); the same closed loop here must pick the same AIME branch every cycle and
    reproduce agent / ego trajectories within 1e-3 m (+ float32 resolution of the ~6.5 km map coordinates) -- in the default
    arithmetic of the predictor (bf16x3) AND with every contraction in fp32, the reference's own precision."""
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    pl, sim, w = make_closed_loop(dict(WORKLOADS[scene]), scripted=False)
    with _arithmetic(pl.network.rt, prec):
        _closed_loop_against_demo_plans(scene, pl, sim, w)


def _closed_loop_against_demo_plans(scene, pl, sim, w):
    D = np.load(os.path.join(ROOT, "tests", "golden", "demo_plans.npz"))
    ulp = float(np.spacing(np.float32(np.abs(w.pos[0, 0]).max())))
    tol = 1e-3 + 2 * ulp
    steps = list(D[scene + "_plan_steps"])
    for pi, step in enumerate(steps):
        while sim.n_plans <= pi:
            planned_at = sim.n_steps
            sim.step()
        assert planned_at == step                                            # same trigger schedule (step 200, 205, ...)
        st, tt = sim.last_result[0][0], sim.last_result[1][0]
        assert len(sim.last_result[0]) == int(D[f"{scene}_p{pi}_n_scen_trees"])
        keys = list(st.nodes.keys())
        assert keys == list(D[f"{scene}_p{pi}_scen_keys"]), (pi, keys)       # identical branch-selection indices
        probs = np.array([float(np.ravel(st.nodes[k].data[0])[0]) for k in keys])
        assert np.abs(probs - D[f"{scene}_p{pi}_scen_probs"]).max() < 1e-5
        for k in keys:
            want = D[f"{scene}_p{pi}_scen_{k}_pos"]
            got = st.nodes[k].data[1][:, ::5]
            assert got.shape == want.shape, (got.shape, want.shape)          # same tracked agents
            assert np.abs(got - want).max() < tol
            assert np.abs(st.nodes[k].data[2][:, ::5] - D[f"{scene}_p{pi}_scen_{k}_cov"]).max() < 1e-3
        tk = [k for k in tt.nodes.keys() if k != -1]
        assert np.array_equal(np.array([tt.nodes[k].parent_key for k in tk]), D[f"{scene}_p{pi}_traj_parent"])
        xs = np.array([tt.nodes[k].data[0] for k in tk])
        assert np.abs(xs[:, :2] - D[f"{scene}_p{pi}_traj_xs"][:, :2]).max() < tol                # ego trajectory [m]
        assert np.abs(xs[:, 2:] - D[f"{scene}_p{pi}_traj_xs"][:, 2:]).max() < 2e-3
        us = np.array([tt.nodes[k].data[1] for k in tk])
        assert np.abs(us - D[f"{scene}_p{pi}_traj_us"]).max() < 2e-3
    while sim.n_steps < steps[-1] + 1:
        sim.step()
    if getattr(sim, "_native", None) is not None:       # (the native loop keeps the observation windows in the library: back into planner.agent_obs)
        sim._native.hand_back()
    assert len(pl.agent_obs) == int(D[scene + "_n_tracked"])
    assert np.abs(np.asarray(sim.ctrl) - D[scene + "_final_ctrl"]).max() < 2e-3
    assert np.abs(sim.state - D[scene + "_final_state"]).max() < tol


@pytest.mark.parametrize("scene", ["demo_1", "demo_2", "demo_3", "demo_4"])
def test_recorded_demo_scenes_branching_weights(scene):
    """The four recorded scenes with the BRANCHING formula weights (mind_amd/weights.py variant "branching"): the reference's
    own closed loop then keeps 2-5 modes per AIME round and runs two rounds per plan (six expansions per plan on demo_1), so
    "identical AIME branch-selection indices on the four demo scenes" is checked on multi-node trees.
    Golden: tests/golden/gen_golden.py demo_branch (the imported reference, CPU, build container), teacher-forced (the ego
    state / control of every cycle are the ones the reference planned from).

    EVERY cycle, strictly: the key lists of ALL scenario trees branch_aime returned, every node id of the internal tree
    with its branch time END_T and end flag, the number of candidate trajectory trees.
    The chosen tree (best_traj_idx = argmin of the candidates' costs) and its ego plan: the reference's tree-iLQR ends in
    poor local minima on some of these cost trees (its own candidate costs range from 0.1 to 22: DESIGN 2 "chaotic cases").
    A cycle where this planner chooses another tree is accepted only if the plan it chose is at least as good under the
    same objective (its cost <= the reference's best cost + 1e-3), i.e. the solver found a better optimum, not a different
    problem; wherever the same tree is chosen, sibling probabilities, agent trajectories and covariances must match, and so must the
    ego plan of the chosen candidate (<= 1e-3 m + float32 resolution of the map coordinates, controls <= 2e-3) -- unless that candidate's fit
    is one where this solver's OWN solution moves by more than the tolerance when its inputs move by their rounding resolution (the
    perturbation probe of the whole-run tests): every miss is classified, an unclassified one fails, and the classified cycles are pinned
    in tests/golden/waived_cycles.json."""
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    from mind_amd.planners.mind.trajectory_tree import flatten_scenario_tree, ilqr_cfg_from
    D = np.load(os.path.join(ROOT, "tests", "golden", "demo_branch.npz"))
    pl, sim, w = make_closed_loop(dict(WORKLOADS[scene]), scripted=False, ckpt="formula_branching:20240121")
    rt, opt = pl.network.rt, pl.traj_tree_opt
    cap = {}
    orig_batch = opt.solve_batch

    def capture(scen_trees, init_state, init_ctrl, target_lane, target_vel):     # what the contingency solves were given
        trees = orig_batch(scen_trees, init_state, init_ctrl, target_lane, target_vel)
        cap["common"] = (ilqr_cfg_from(opt.config, "w_opt_cfg"), ilqr_cfg_from(opt.config, "opt_cfg"),
                         opt._get_init_state(init_state, init_ctrl), np.asarray(target_lane, np.float64), target_vel)
        cap["scen_trees"], cap["xs"] = scen_trees, [t._arrays[0][1:] for t in trees]
        return trees

    def ill_conditioned(j):
        cw, cf, x0, lane, tv = cap["common"]
        return _solution_moves_under_rounding_noise(rt.ilqr_contingency, (cw, cf, [flatten_scenario_tree(cap["scen_trees"][j])], x0, lane, tv), cap["xs"][j])

    opt.solve_batch = capture
    ulp = float(np.spacing(np.float32(np.abs(w.pos[0, 0]).max())))
    tol = 1e-3 + 2 * ulp
    steps = list(D[scene + "_plan_steps"])
    state_in, ctrl_in = D[scene + "_state_in"], D[scene + "_ctrl_in"]
    same_choice = ego_ok = 0
    better, other_ill, ego_ill = [], [], []
    for pi, step in enumerate(steps):
        while sim.n_plans <= pi:
            will_plan = sim.sim_time >= sim.enable_time and (sim.last_trigger is None or
                                                             sim.sim_time - sim.last_trigger >= sim.PLAN_STEP)
            if will_plan and sim.enabled:
                sim.state, sim.ctrl = state_in[pi].copy(), ctrl_in[pi].copy()
            planned_at = sim.n_steps
            sim.step()
        assert planned_at == step
        gen = pl.scen_tree_gen
        # ---- the whole AIME result (discrete: must be identical)
        trees = gen.get_scenario_tree()
        all_trees = ["|".join(t.nodes.keys()) for t in trees]
        assert all_trees == list(D[f"{scene}_p{pi}_all_tree_keys"]), (pi, all_trees)
        nodes = sorted((k, int(n.data.data["END_T"]), bool(n.data.end_flag)) for k, n in gen.tree.nodes.items() if k != "root")
        assert [n[0] for n in nodes] == list(D[f"{scene}_p{pi}_all_node_ids"])
        assert [n[1] for n in nodes] == list(D[f"{scene}_p{pi}_all_node_end_t"])
        assert [n[2] for n in nodes] == list(D[f"{scene}_p{pi}_all_node_end_flag"])
        assert len(nodes) > 1                               # every cycle really branched
        ref_costs, costs = D[f"{scene}_p{pi}_tree_costs"], np.array(pl.timing["tree_costs"])
        assert len(costs) == len(ref_costs) == len(all_trees)
        # ---- the chosen tree
        st, tt = sim.last_result[0][0], sim.last_result[1][0]
        keys = list(st.nodes.keys())
        if keys != list(D[f"{scene}_p{pi}_scen_keys"]):
            if costs.min() <= ref_costs.min() + 1e-3:              # a better optimum, not a different problem
                better.append(pi)
            else:                                                  # ... or the candidate the two solvers disagree on is ill-conditioned
                j = int(np.argmax(np.abs(costs - ref_costs)))
                moved = ill_conditioned(j)
                assert moved > tol, (pi, keys, costs, ref_costs, moved)
                other_ill.append(pi)
            continue
        same_choice += 1
        probs = np.array([float(np.ravel(st.nodes[k].data[0])[0]) for k in keys])
        assert np.abs(probs - D[f"{scene}_p{pi}_scen_probs"]).max() < 1e-5
        for k in keys:
            want = D[f"{scene}_p{pi}_scen_{k}_pos"]
            got = st.nodes[k].data[1][:, ::5]
            assert got.shape == want.shape, (got.shape, want.shape)
            assert np.abs(got - want).max() < tol
            assert np.abs(st.nodes[k].data[2][:, ::5] - D[f"{scene}_p{pi}_scen_{k}_cov"]).max() < 1e-3
        tk = [k for k in tt.nodes.keys() if k != -1]
        assert np.array_equal(np.array([tt.nodes[k].parent_key for k in tk]), D[f"{scene}_p{pi}_traj_parent"])
        best = int(np.argmin(ref_costs))
        xs = np.array([tt.nodes[k].data[0] for k in tk])
        us = np.array([tt.nodes[k].data[1] for k in tk])
        d_xy = float(np.abs(xs[:, :2] - D[f"{scene}_p{pi}_traj_xs"][:, :2]).max())
        d_u = float(np.abs(us - D[f"{scene}_p{pi}_traj_us"]).max())
        if d_xy < tol and d_u < 2e-3:
            ego_ok += 1
            continue
        # the same tree with another ego plan: only where the chosen candidate's fit is ill-conditioned under its own inputs' noise
        moved = ill_conditioned(int(pl.timing["best_traj_idx"]))
        print(f"{scene} cycle {pi}: same tree {keys[0]}, ego plan deviates by {d_xy:.2e} m / {d_u:.2e} (controls); costs {costs[best]:.6f} vs the "
              f"reference's {ref_costs[best]:.6f}; own solution moves by {moved:.2e} m under rounding noise")
        assert moved > tol, (pi, d_xy, d_u, moved, costs, ref_costs)         # a well-conditioned fit that disagrees is a real failure
        ego_ill.append(pi)
    print(f"{scene}: chosen tree equal in {same_choice}/{len(steps)} cycles (better optimum in {better}, ill-conditioned other choice in {other_ill}), "
          f"ego plan within tolerance in {ego_ok}, ill-conditioned chosen fit in {ego_ill}")
    _check_waived("branching_weights_12_cycles", scene, {"better_optimum": better, "other_choice_ill_conditioned": other_ill, "ego_plan_ill_conditioned": ego_ill})


@pytest.mark.parametrize("prec", ["bf16x3", "f32", "bf16x6"])
@pytest.mark.parametrize("scene", ["demo_1", "demo_2", "demo_3", "demo_4"])
def test_recorded_demo_scenes_branching_weights_whole_run(scene, prec):
    """The reference's WHOLE closed loop on the recorded scenes (t = 4.0 .. 9.9 s, 60 planning cycles) with the branching formula
    weights, teacher-forced (tests/golden/gen_golden.py demo_branch_runs: discrete results only).  In EVERY cycle the AIME
    result must be identical: the key lists of all scenario trees, every internal node's id, branch time END_T and end flag,
    the number of candidate trajectory trees.
    The chosen tree is the argmin over the candidates' tree-iLQR costs, and the reference's tree-iLQR is ill-conditioned on some
    cost trees (DESIGN 2 "chaotic cases": either solver can stop in a poor local minimum -- candidate costs of 1e4 next to 0.2).
    A cycle with another choice is therefore accepted only if this planner's best cost is at least as good as the reference's,
    or if the candidate whose cost disagrees is one where this solver's OWN answer moves by more than the parity tolerance
    when its inputs are perturbed by their rounding resolution (the criterion of the plain-weights whole-run test).  The cycles of
    either kind are pinned (tests/golden/waived_cycles.json): none can silently join or leave them.
    Both arithmetics of the predictor: the default (bf16x3) and fp32 throughout, the reference's own precision."""
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    pl, sim, w = make_closed_loop(dict(WORKLOADS[scene]), scripted=False, ckpt="formula_branching:20240121")
    with _arithmetic(pl.network.rt, prec):
        _whole_run_against_demo_branch_runs(scene, prec, pl, sim, w)


def _whole_run_against_demo_branch_runs(scene, prec, pl, sim, w):
    from mind_amd.planners.mind.trajectory_tree import flatten_scenario_tree, ilqr_cfg_from
    D = np.load(os.path.join(ROOT, "tests", "golden", "demo_branch_runs.npz"))
    rt, opt = pl.network.rt, pl.traj_tree_opt
    cap = {}
    orig_batch = opt.solve_batch

    def capture(scen_trees, init_state, init_ctrl, target_lane, target_vel):     # what the contingency solves were given
        trees = orig_batch(scen_trees, init_state, init_ctrl, target_lane, target_vel)
        cap["common"] = (ilqr_cfg_from(opt.config, "w_opt_cfg"), ilqr_cfg_from(opt.config, "opt_cfg"),
                         opt._get_init_state(init_state, init_ctrl), np.asarray(target_lane, np.float64), target_vel)
        cap["scen_trees"], cap["xs"] = scen_trees, [t._arrays[0][1:] for t in trees]
        return trees

    opt.solve_batch = capture
    ulp = float(np.spacing(np.float32(np.abs(w.pos[0, 0]).max())))
    tol = 1e-3 + 2 * ulp
    steps = list(D[scene + "_plan_steps"])
    state_in, ctrl_in = D[scene + "_state_in"], D[scene + "_ctrl_in"]
    assert len(steps) == 60
    same_choice, better, ill, n_nodes, margins = 0, [], [], [], []
    for pi, step in enumerate(steps):
        while sim.n_plans <= pi:
            will_plan = sim.sim_time >= sim.enable_time and (sim.last_trigger is None or
                                                             sim.sim_time - sim.last_trigger >= sim.PLAN_STEP)
            if will_plan and sim.enabled:
                sim.state, sim.ctrl = state_in[pi].copy(), ctrl_in[pi].copy()
            planned_at = sim.n_steps
            sim.step()
        assert planned_at == step
        gen = pl.scen_tree_gen
        all_trees = ["|".join(t.nodes.keys()) for t in gen.get_scenario_tree()]
        assert all_trees == list(D[f"{scene}_p{pi}_all_tree_keys"]), (pi, all_trees)
        nodes = sorted((k, int(n.data.data["END_T"]), bool(n.data.end_flag)) for k, n in gen.tree.nodes.items() if k != "root")
        assert [n[0] for n in nodes] == list(D[f"{scene}_p{pi}_all_node_ids"]), pi
        assert [n[1] for n in nodes] == list(D[f"{scene}_p{pi}_all_node_end_t"]), pi
        assert [n[2] for n in nodes] == list(D[f"{scene}_p{pi}_all_node_end_flag"]), pi
        n_nodes.append(len(nodes))
        ref_costs, costs = D[f"{scene}_p{pi}_tree_costs"], np.array(pl.timing["tree_costs"])
        assert len(costs) == len(ref_costs) == len(all_trees)
        keys = list(sim.last_result[0][0].nodes.keys())
        if keys == list(D[f"{scene}_p{pi}_scen_keys"]):
            same_choice += 1
            continue
        # cost margins of a cycle with another choice: this planner's cost of its own choice and of the reference's, the reference's of both
        ri, oi_ = int(np.argmin(ref_costs)), int(np.argmin(costs))
        margins.append((pi, round(float(costs[oi_]), 4), round(float(costs[ri]), 4), round(float(ref_costs[ri]), 4), round(float(ref_costs[oi_]), 4)))
        if costs.min() <= ref_costs.min() + 1e-3:
            better.append(pi)
            continue
        j = int(np.argmax(np.abs(costs - ref_costs)))                       # the candidate the two solvers disagree on
        cw, cf, x0, lane, tv = cap["common"]
        moved = _solution_moves_under_rounding_noise(rt.ilqr_contingency, (cw, cf, [flatten_scenario_tree(cap["scen_trees"][j])], x0, lane, tv),
                                                     cap["xs"][j])
        assert moved > tol, (pi, j, costs, ref_costs, moved)                 # a well-conditioned candidate that disagrees is a real failure
        ill.append(pi)
    print(f"{scene} [{prec}]: 60/60 cycles with the reference's AIME tree ({min(n_nodes)}..{max(n_nodes)} nodes); same tree chosen in "
          f"{same_choice}, better optimum in {better}, ill-conditioned candidate in {ill}; (cycle, own cost of own / of the reference's choice, "
          f"reference's cost of its own / of this planner's choice): {margins}")
    # every cycle with another choice is classified above; WHICH cycles are is pinned (tests/golden/waived_cycles.json: fp32 59 / 54 / 60 / 60 of
    # 60 cycles choose the reference's tree outright = 233 of 240, bf16x3 59 / 53 / 60 / 60 = 232; the same sets on every box and build since
    # round 4 -- the kernels are deterministic, the chaotic fits move only when somebody changes last bits)
    assert max(n_nodes) > 1 and same_choice == 60 - len(better) - len(ill)
    _check_waived("branching_weights_whole_run", f"{scene}/{prec}", {"better_optimum": better, "ill_conditioned": ill})


def test_full_tree_plan_is_identical_with_device_assembled_windows():
    """The full scripted 6-ary tree (rounds of 1 / 6 / 36 / 216 scenes): from the third round on, the re-basing windows of the kept
    modes are cut out of the previous round's device arena and the device-resident rows (mind_aime_rebase device source) instead
    of being stacked and uploaded.  The plan -- every scenario tree's nodes with probabilities, agent trajectories and
    covariances, the ego plan, the control -- must be bit-identical to the run that uploads the windows, and the lazily held
    host windows must equal the uploaded ones when somebody asks for them."""
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    from mind_amd.planners.mind.scenario_tree import DevScene
    res, lazies = [], []
    for dev_win in (True, False):
        pl, sim, w = make_closed_loop(dict(WORKLOADS["cfg4tree"]), full_tree=True, speculative=False)
        pl.scen_tree_gen.device_windows = dev_win
        pl.scen_tree_gen.native_aime = False      # the round-by-round path is what has the two window sources (mind_aime_plan: always device)
        sim.run_plans(2)
        trees = pl.scen_tree_gen.get_scenario_tree()
        flat = []
        for t in trees:
            for k, n in t.nodes.items():
                flat.append((k, float(np.ravel(n.data[0])[0]), np.asarray(n.data[1]).copy(), np.asarray(n.data[2]).copy()))
        tt = sim.last_result[1][0]
        res.append((flat, np.array(sim.ctrl), tt._arrays[0].copy(), pl.timing["best_traj_idx"]))
        obs = [n.data.obs_data for n in pl.scen_tree_gen.tree.nodes.values() if isinstance(getattr(n.data, "obs_data", None), DevScene)]
        lazies.append((sum(1 for o_ in obs if o_.lazy is not None), {o_["SCEN_ID"]: o_ for o_ in obs}))
    (fa, ca, xa, ba), (fb, cb, xb, bb) = res
    assert len(fa) == len(fb) > 200 and ba == bb
    for (ka, pa, ma, va), (kb, pb, mb, vb) in zip(fa, fb):
        assert ka == kb and pa == pb and np.array_equal(ma, mb) and np.array_equal(va, vb), ka
    assert np.array_equal(ca, cb) and np.array_equal(xa, xb)
    assert lazies[0][0] >= 36 and lazies[1][0] == 0                      # rounds 3 and 4 really took the device path
    for sid, o_ in list(lazies[0][1].items())[::17]:                      # lazily materialised host windows = the uploaded ones
        ref = lazies[1][1][sid]
        for key in ("TRAJS_POS_HIST", "TRAJS_ANG_HIST", "TRAJS_VEL_HIST", "TRAJS_COV_HIST"):
            assert np.array_equal(o_[key], ref[key]), (sid, key)


def test_checkpoint_tar_goes_through_the_same_loader(tmp_path):
    """The reference's checkpoint format (planner.py:46-47: torch.load(path)["state_dict"], a .tar written by torch.save):
    a planner configured with such a file plans exactly what the formula-weight planner plans."""
    import json
    import torch
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    from mind_amd.weights import formula_state_dict
    ck = os.path.join(tmp_path, "20240121-172745.tar")
    torch.save({"state_dict": formula_state_dict(as_torch=True), "epoch": 0}, ck)
    outs = []
    for ckpt in (None, ck):
        pl, sim, w = make_closed_loop(dict(WORKLOADS["demo_3"]), scripted=False, ckpt=ckpt)
        sim.run_plans(2)
        st, tt = sim.last_result[0][0], sim.last_result[1][0]
        outs.append((list(st.nodes.keys()), next(iter(st.nodes.values())).data[1].copy(), np.array(sim.ctrl), np.array(sim.state)))
    assert outs[0][0] == outs[1][0]
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("ckpt", [None, "formula_branching:20240121"])
def test_four_recorded_scenes_fused_in_one_process_equal_separate_runs(ckpt):
    """BASELINE config 3 as written: demo_1..4 planned concurrently by ONE process, the AIME rounds of all four scenes merged
    into one predictor batch per round (mind_amd.fused).  Every scene must plan exactly what it plans alone -- same branch
    ids, bit-identical trajectories, controls and ego states over three cycles -- with the plain formula weights (one round
    per plan) and with the branching weights (two rounds, 2-6 expansions per scene and plan)."""
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    from mind_amd.fused import FusedClosedLoops
    scenes = ["demo_1", "demo_2", "demo_3", "demo_4"]

    def snapshot(pl, sim):
        st = sim.last_result[0][0]
        return (list(st.nodes.keys()), [t_.nodes.keys().__len__() for t_ in pl.scen_tree_gen.get_scenario_tree()],
                np.concatenate([st.nodes[k].data[1].ravel() for k in st.nodes]), np.array(sim.ctrl), np.array(sim.state),
                pl.timing["best_traj_idx"])

    alone = []
    for sc in scenes:
        pl, sim, w = make_closed_loop(dict(WORKLOADS[sc]), scripted=False, speculative=False, ckpt=ckpt)
        # the fused rounds are fed by the host featuriser; a scene planned alone builds its root on the device by default, which agrees
        # with the host featuriser to float32 rounding only (tests/test_gpu_aime_native.py): same featuriser on both sides here
        pl.scen_tree_gen.device_root = False
        snaps = []
        for _ in range(3):
            sim.run_plans(1)
            snaps.append(snapshot(pl, sim))
        alone.append((snaps, pl.scen_tree_gen.n_expanded))
    loops = [make_closed_loop(dict(WORKLOADS[sc]), scripted=False, speculative=False, ckpt=ckpt) for sc in scenes]
    fl = FusedClosedLoops([l[1] for l in loops])
    for c in range(3):
        fl.run_plans(1)
        for i, (pl, sim, w) in enumerate(loops):
            a, b = alone[i][0][c], snapshot(pl, sim)
            assert a[0] == b[0] and a[1] == b[1] and a[5] == b[5], (scenes[i], c)
            for x, y in zip(a[2:5], b[2:5]):
                assert np.array_equal(x, y), (scenes[i], c)
    assert [l[0].scen_tree_gen.n_expanded for l in loops] == [a[1] for a in alone]
    # the rounds really were merged: four scenes per predictor call in the root round
    assert fl.fused.n_scenes > fl.fused.n_calls and fl.fused.n_scenes == sum(a[1] for a in alone)
    if ckpt is not None:
        assert fl.fused.n_scenes >= 3 * (4 + 4)         # branching weights: at least a second round per scene and plan


def test_four_recorded_scenes_pipelined_from_one_thread_equal_separate_runs():
    """BASELINE config 3 from ONE host thread (mind_amd.pipelined): every planner on its own HIP context / stream, scene i + 1's
    plan_begin (AIME rounds, start of the contingency solves) before scene i's plan_end, so that one scene's tree-iLQR runs on the device
    beside the next scene's predictor.  The scenes stay independent closed loops: over four cycles every scene plans bit for bit what it
    plans alone (branch ids, trajectories, controls, ego states), the native AIME plan and the solves started ahead included."""
    sys.path.insert(0, ROOT)
    from bench import BRANCHING_WEIGHTS, WORKLOADS, make_closed_loop
    from mind_amd.pipelined import PipelinedClosedLoops
    scenes = ["demo_1", "demo_2", "demo_3", "demo_4"]

    def snapshot(pl, sim):
        st = sim.last_result[0][0]
        return (list(st.nodes.keys()), np.concatenate([st.nodes[k].data[1].ravel() for k in st.nodes]), np.array(sim.ctrl), np.array(sim.state),
                pl.timing["best_traj_idx"], sim.n_steps)

    alone = []
    for sc in scenes:
        pl, sim, w = make_closed_loop(dict(WORKLOADS[sc]), scripted=False, speculative=False, ckpt=BRANCHING_WEIGHTS)
        snaps = []
        for _ in range(4):
            sim.run_plans(1)
            snaps.append(snapshot(pl, sim))
        alone.append(snaps)
    loops = [make_closed_loop(dict(WORKLOADS[sc]), scripted=False, speculative=False, ckpt=BRANCHING_WEIGHTS, own_context=True) for sc in scenes]
    assert len({id(l[0].network.rt) for l in loops}) == 4 and len({l[0].network.rt.ctx.value for l in loops}) == 4      # four contexts
    pc = PipelinedClosedLoops([l[1] for l in loops])
    for c in range(4):
        pc.run_plans(1)
        for i, (pl, sim, w) in enumerate(loops):
            a, b = alone[i][c], snapshot(pl, sim)
            assert a[0] == b[0] and a[4] == b[4] and a[5] == b[5], (scenes[i], c)
            for x, y in zip(a[1:4], b[1:4]):
                assert np.array_equal(x, y), (scenes[i], c)
    assert all(l[0].scen_tree_gen.n_native_plans == 4 for l in loops)
    for pl, sim, w in loops:
        pl.network.rt.close()


@pytest.mark.parametrize("scene", ["demo_1", "demo_4"])
def test_aime_tree_does_not_depend_on_the_pair_kernel_arithmetic(scene):
    """The discrete AIME result (node ids, branch times, chosen tree) of a branching closed loop is the same whether the pair
    kernel runs on the fp32 MFMA or in its default bf16x3 arithmetic, and the trajectories agree to 1e-4 m: the decisions do
    not sit on the rounding difference between the two fp32-class modes."""
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    runs = {}
    for prec in ("f32", "bf16x3"):
        pl, sim, w = make_closed_loop(dict(WORKLOADS[scene]), scripted=False, speculative=False, ckpt="formula_branching:20240121")
        before = pl.network.rt.pair_precision()
        try:
            pl.network.rt.set_pair_precision(prec)
            snaps = []
            for _ in range(3):
                sim.run_plans(1)
                gen = pl.scen_tree_gen
                nodes = sorted((k, int(n.data.data["END_T"]), bool(n.data.end_flag)) for k, n in gen.tree.nodes.items() if k != "root")
                st = sim.last_result[0][0]
                snaps.append((nodes, list(st.nodes.keys()), np.concatenate([st.nodes[k].data[1].ravel() for k in st.nodes]),
                              np.array(sim.state)))
            runs[prec] = snaps
        finally:
            pl.network.rt.set_pair_precision(before)
    compared = 0
    for a, b in zip(runs["f32"], runs["bf16x3"]):
        assert a[0] == b[0]                                 # every AIME node, its branch time and end flag
        if a[1] != b[1]:
            # the candidate choice of this cycle flipped: an ill-conditioned tree-iLQR solve (DESIGN 2 "chaotic cases", also
            # seen between the reference and itself); the two free-running loops are different runs from here on
            break
        assert np.abs(a[2] - b[2]).max() < 1e-4 + 2 * float(np.spacing(np.float32(np.abs(a[2]).max())))
        assert np.abs(a[3] - b[3]).max() < 1e-3
        compared += 1
    assert compared >= 2


def _solution_moves_under_rounding_noise(solve, args, xs_ref):
    """Largest displacement of the ego trajectory when the solver inputs are moved by their own rounding resolution:
    agent means by +-1 float32 ulp (4 draws), the initial state by a relative 1e-13 (2 draws)."""
    cw, cf, flats, x0, lane, tv = args
    moved = 0.0
    for seed in range(6):
        rng = np.random.default_rng(seed)
        f2, x2 = dict(flats[0]), np.array(x0, dtype=np.float64)
        if seed < 4:
            m = flats[0]["mean"]
            f2["mean"] = np.where(rng.random(m.shape) < 0.5, np.nextafter(m, np.float32(np.inf)),
                                  np.nextafter(m, np.float32(-np.inf))).astype(np.float32)
        else:
            x2 = x2 * (1.0 + 1e-13 * rng.standard_normal(x2.shape))
        px = solve(cw, cf, [f2], x2, lane, tv)[0]
        moved = max(moved, float(np.abs(px[0][:, :2] - xs_ref[:, :2]).max()))
    return moved


@pytest.mark.parametrize("scene", ["demo_1", "demo_2", "demo_3", "demo_4"])
def test_recorded_demo_scenes_whole_run_teacher_forced(scene):
    """Every planning cycle of the reference's whole closed loop on the recorded scenes (t = 4.0 .. 9.9 s, 60 cycles,
    tests/golden/gen_golden.py demo_runs), teacher-forced: before each cycle the ego state / control are set to the ones the
    reference planned from (free-running loops drift apart once a discrete decision flips).
    Compared in EVERY cycle: number of scenario trees, AIME branch ids, tracked agents, root probability, agent
    trajectories and covariances (every 4th cycle is stored).
    Ego trajectory / control: the reference's tree-iLQR is ill-conditioned on some of these cost trees (Python round()
    cell lookups and an LM schedule amplify 1e-16 differences into different local minima, DESIGN 2 "chaotic cases").
    A cycle whose ego plan differs by more than the tolerance must therefore be one where this solver's OWN answer moves
    by more than the tolerance when its inputs are perturbed by their rounding resolution (+-1 float32 ulp of the agent
    means, 1e-13 relative on the initial state); otherwise the test fails.  The ill-conditioned cycles are pinned
    (tests/golden/waived_cycles.json): none can silently join or leave them."""
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    D = np.load(os.path.join(ROOT, "tests", "golden", "demo_runs.npz"))
    pl, sim, w = make_closed_loop(dict(WORKLOADS[scene]), scripted=False)
    from mind_amd.planners.mind.trajectory_tree import flatten_scenario_tree, ilqr_cfg_from
    rt, opt = pl.network.rt, pl.traj_tree_opt
    cap = {}
    orig_batch = opt.solve_batch

    def capture(scen_trees, init_state, init_ctrl, target_lane, target_vel):     # what the contingency solves were given
        trees = orig_batch(scen_trees, init_state, init_ctrl, target_lane, target_vel)
        cap["args"] = (ilqr_cfg_from(opt.config, "w_opt_cfg"), ilqr_cfg_from(opt.config, "opt_cfg"),
                       [flatten_scenario_tree(scen_trees[0])], opt._get_init_state(init_state, init_ctrl),
                       np.asarray(target_lane, np.float64), target_vel)
        cap["xs"] = trees[0]._arrays[0][1:]
        return trees

    opt.solve_batch = capture
    ulp = float(np.spacing(np.float32(np.abs(w.pos[0, 0]).max())))
    tol = 1e-3 + 2 * ulp
    state_in, ctrl_in, ctrl_out = D[scene + "_state_in"], D[scene + "_ctrl_in"], D[scene + "_ctrl_out"]
    keys, n_agents, xs_all = D[scene + "_scen_keys"], D[scene + "_n_agents"], D[scene + "_traj_xs"]
    n = len(state_in)
    assert n == 60
    agree, ill = 0, []
    worst_agents = 0.0
    for pi in range(n):
        while True:
            will_plan = sim.sim_time >= sim.enable_time and (sim.last_trigger is None or
                                                             sim.sim_time - sim.last_trigger >= sim.PLAN_STEP)
            if will_plan and pi > 0:
                sim.state, sim.ctrl = state_in[pi].copy(), ctrl_in[pi].copy()
            if sim.step():
                break
        if pi == 0:
            assert np.array_equal(pl.state, state_in[0])                      # the recording itself: identical by construction
        st, tt = sim.last_result[0][0], sim.last_result[1][0]
        assert len(sim.last_result[0]) == int(D[scene + "_n_scen_trees"][pi]), pi
        assert "|".join(st.nodes.keys()) == str(keys[pi]), (pi, list(st.nodes.keys()), keys[pi])
        root = next(iter(st.nodes.values())).data
        assert root[1].shape[0] == int(n_agents[pi]), pi
        assert abs(float(np.ravel(root[0])[0]) - float(D[scene + "_root_prob"][pi])) < 1e-5
        if pi % 4 == 0:                                                       # agent trajectories: every 4th cycle is stored
            worst_agents = max(worst_agents, float(np.abs(root[1][:, ::10] - D[f"{scene}_p{pi}_pos"]).max()))
            assert np.abs(root[2][:, ::10] - D[f"{scene}_p{pi}_cov"]).max() < 1e-3
        tk = [k for k in tt.nodes.keys() if k != -1]
        xs = np.array([tt.nodes[k].data[0] for k in tk])[:25]
        d_ego = float(np.abs(xs[:, :2] - xs_all[pi][:, :2]).max())
        d_ctrl = float(np.abs(np.asarray(sim.ctrl) - ctrl_out[pi]).max())
        if d_ego < tol and d_ctrl < 2e-3:
            agree += 1
            continue
        moved = _solution_moves_under_rounding_noise(rt.ilqr_contingency, cap["args"], cap["xs"])
        assert moved > tol, (pi, d_ego, d_ctrl, moved)       # a well-conditioned cycle that disagrees is a real failure
        ill.append(pi)
    assert worst_agents < tol, worst_agents
    # every cycle that disagrees is shown ill-conditioned above; WHICH cycles do is pinned (tests/golden/waived_cycles.json: 60 / 49 / 58 / 56 of
    # 60 agree outright = 223 of 240; the chaotic cycles move only with the last bits of their inputs)
    assert agree == n - len(ill)
    _check_waived("plain_weights_whole_run", scene, {"ill_conditioned": ill})
    # these scenes grow one chain-shaped tree every cycle: from the second cycle on the warm-start fit is the one that ran
    # beside the predictor (speculate_warm) -- the comparisons above therefore cover that path
    assert opt.counters["warm_hits"] >= n - 2, opt.counters
    print(f"[{scene}] {agree}/{n} cycles agree outright; ill-conditioned cycles: {ill}")


@pytest.mark.parametrize("workload", ["demo1", "demo_1"])
def test_speculative_warm_start_is_bit_identical(workload):
    """The warm-start fits that run beside the predictor (previous cycle's tree shapes, second stream, own HIP context)
    give exactly the controls the in-line path computes: same closed loop with and without speculation."""
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    runs = {}
    for spec in (True, False):
        pl, sim, w = make_closed_loop(dict(WORKLOADS[workload]))
        pl.traj_tree_opt.speculative = spec
        out = []
        for _ in range(6):
            sim.run_plans(1)
            tt = sim.last_result[1][0]
            out.append((np.array(sim.ctrl), tt._arrays[0].copy(), tt._arrays[1].copy(), pl.timing["best_traj_idx"]))
        runs[spec] = (out, dict(pl.traj_tree_opt.counters))
    for (c1, x1, u1, b1), (c2, x2, u2, b2) in zip(runs[True][0], runs[False][0]):
        assert b1 == b2 and np.array_equal(c1, c2) and np.array_equal(x1, x2) and np.array_equal(u1, u2)
    on, off = runs[True][1], runs[False][1]
    assert off["warm_speculated"] == 0 and off["warm_hits"] == 0
    assert on["warm_hits"] >= 3 and on["solves"] == off["solves"] and on["iterations"] == off["iterations"]


class _MovedWorld:
    """A SynthWorld seen from a frame rotated by `th` and shifted by `t` (a rigid motion of the whole scene)."""

    def __init__(self, base, th, t):
        self.b, self.th = base, th
        self.R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        self.t = np.asarray(t, dtype=np.float64)
        self.n_agents, self.agent_ids = base.n_agents, base.agent_ids
        self.target_lane = self.fwd(np.asarray(base.target_lane, np.float64)).astype(np.float32)
        self.target_lane_info, self.target_velocity = base.target_lane_info, base.target_velocity
        self.vector_lane_segments = base.vector_lane_segments
        self.agent_speed = base.agent_speed

    def fwd(self, p):
        return p @ self.R.T + self.t

    def inv(self, p):
        return (p - self.t) @ self.R

    def get_lane_segment_centerline(self, lane_id):
        c = np.array(self.b.get_lane_segment_centerline(lane_id), dtype=np.float64)
        c[:, :2] = self.fwd(c[:, :2])
        return c

    def agent_state(self, i, t):
        s = self.b.agent_state(i, t)
        xy = self.fwd(s[None, :2])[0]
        return np.array([xy[0], xy[1], s[2], s[3] + self.th])

    def object_type(self, i):
        return self.b.object_type(i)


def test_planning_cycle_is_equivariant_under_rigid_motion():
    """Size-independent property of the whole path (featurisation -> predictor -> AIME -> world frame -> tree-iLQR): moving
    the entire scene rigidly (rotation 0.7 rad, shift (310, -95) m) moves every planned quantity with it -- same branch
    ids and probabilities, agent trajectories and the ego plan equal after mapping back (float32 resolution of the larger
    coordinates; the first two tree-iLQR iterations only: later ones amplify that rounding)."""
    sys.path.insert(0, ROOT)
    from mind_amd.closed_loop import ClosedLoopSim
    from mind_amd.planners.mind.planner import MINDPlanner
    from mind_amd.synth import ScriptedBranching
    res = {}
    base = SynthWorld(n_agents=12, n_lanes=3, n_segs=10, seed=2)
    for name, w in (("ref", base), ("moved", _MovedWorld(base, 0.7, (310.0, -95.0)))):
        pl = MINDPlanner(CFG)
        pl.scen_tree_gen.network = ScriptedBranching(pl.network)
        sim = ClosedLoopSim(w, pl)
        sim.run_until(4.0)
        import mind_amd.planners.mind.trajectory_tree as TT
        orig = TT.ilqr_cfg_from
        TT.ilqr_cfg_from = lambda config, block, max_iter=100: orig(config, block, max_iter=2)
        try:
            sim.run_plans(1)
        finally:
            TT.ilqr_cfg_from = orig
        trees = pl.scen_tree_gen.get_scenario_tree()
        res[name] = (w, trees, sim.last_result[1][0]._arrays[0].copy(), pl.timing["best_traj_idx"])
    (w0, t0, x0, b0), (w1, t1, x1, b1) = res["ref"], res["moved"]
    assert b0 == b1 and len(t0) == len(t1) >= 2
    for a, b in zip(t0, t1):
        assert list(a.nodes.keys()) == list(b.nodes.keys())                       # same branch ids and tree shapes
        for k in a.nodes:
            da, db = a.nodes[k].data, b.nodes[k].data
            assert abs(float(np.ravel(da[0])[0]) - float(np.ravel(db[0])[0])) < 1e-5
            back = w1.inv(db[1].astype(np.float64).reshape(-1, 2)).reshape(db[1].shape)
            assert np.abs(back - da[1]).max() < 2e-3                              # agent trajectories [m]
            assert np.abs(da[2] - db[2]).max() < 1e-4                             # sigmas are frame independent
    ego_back = w1.inv(x1[:, :2])
    assert np.abs(ego_back - x0[:, :2]).max() < 2e-2                              # ego plan [m] after two iterations
    assert np.abs(x1[:, 2] - x0[:, 2]).max() < 2e-2                               # speeds
    d_yaw = np.arctan2(np.sin(x1[:, 3] - 0.7 - x0[:, 3]), np.cos(x1[:, 3] - 0.7 - x0[:, 3]))
    assert np.abs(d_yaw).max() < 5e-3


def _first_divergence(hip, ref):
    """hip [n,4] (mind_last_ilqr_trace rows), ref [m,3] (gen_golden.py demo_traces rows: mu, J_opt, outcome 1 / -1 / -2): the first
    iteration at which the two fits part -- another line-search outcome, another mu, or J of the nominal trajectory off by more than
    1e-4 relative (the inputs already differ by the predictors' float32 noise) -- or None."""
    out_h = np.where(hip[:, 2] >= 0, 1.0, hip[:, 2])
    for i in range(max(len(hip), len(ref))):
        if i >= len(hip) or i >= len(ref):
            return i
        if out_h[i] != ref[i, 2] or abs(hip[i, 0] - ref[i, 0]) > 1e-9 * abs(ref[i, 0]) or abs(hip[i, 1] - ref[i, 1]) > 1e-4 * abs(ref[i, 1]):
            return i
    return None


@pytest.mark.parametrize("variant", ["plain", "branching"])
@pytest.mark.parametrize("scene", ["demo_1", "demo_2", "demo_3", "demo_4"])
def test_tree_ilqr_follows_the_reference_iteration_by_iteration(scene, variant):
    """iLQR.fit's loop (planners/ilqr/solver.py:133-158) compared ITERATION BY ITERATION with the reference's own, over its whole
    closed loop on the recorded scenes (60 teacher-forced planning cycles, every candidate scenario tree, warm-start and full fit;
    tests/golden/gen_golden.py demo_traces / demo_branch_traces): the Levenberg-Marquardt value of every backward pass, the cost of the
    nominal trajectory when the line search starts and the outcome (accepted / all ten step sizes rejected / singular Q_uu), from
    mind_last_ilqr_trace.
    A fit may leave the reference's trace only where the reference leaves ITS OWN trace when its inputs move by their rounding
    resolution (golden `split`: ego state x (1 + 1e-13), predicted means +-1 float32 ulp; the reference's solver amplifies such noise
    on some cost trees -- DESIGN 2 "chaotic cases"), where this solver leaves its own trace under the same noise (the golden holds two
    perturbed reference runs per fit, and none on the node probabilities), or behind a warm-start fit that did (the full fit starts
    from its controls).
    This is the iteration-level form of the best_traj_idx / ego-plan comparison of the whole-run tests above."""
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    branching = variant == "branching"
    G = np.load(os.path.join(ROOT, "tests", "golden", "demo_branch_traces.npz" if branching else "demo_traces.npz"))
    D = np.load(os.path.join(ROOT, "tests", "golden", "demo_branch_runs.npz" if branching else "demo_runs.npz"))
    rows, index, split = G[scene + "_trace_rows"], G[scene + "_trace_index"], G[scene + "_trace_split"]
    fits = {(int(i[0]), int(i[1]), int(i[2])): (rows[i[3]:i[3] + i[4]], int(s)) for i, s in zip(index, split)}
    pl, sim, w = make_closed_loop(dict(WORKLOADS[scene]), scripted=False, ckpt="formula_branching:20240121" if branching else None)
    from mind_amd.planners.mind.trajectory_tree import flatten_scenario_tree, ilqr_cfg_from
    rt, opt = pl.network.rt, pl.traj_tree_opt
    opt.speculative = False                       # both fits of every candidate in one launch on this context
    got = {}
    orig_batch = opt.solve_batch

    def capture(scen_trees, init_state, init_ctrl, target_lane, target_vel):
        trees = orig_batch(scen_trees, init_state, init_ctrl, target_lane, target_vel)
        got["traces"] = [(rt.ilqr_trace(t, 0), rt.ilqr_trace(t, 1)) for t in range(len(scen_trees))]
        got["args"] = (ilqr_cfg_from(opt.config, "w_opt_cfg"), ilqr_cfg_from(opt.config, "opt_cfg"),
                       [getattr(t, "_flat", None) or flatten_scenario_tree(t) for t in scen_trees], opt._get_init_state(init_state, init_ctrl),
                       np.asarray(target_lane, np.float64), target_vel)
        return trees

    def own_split(ti, ph, base):
        """first iteration at which THIS solver leaves its own trace when the fit's inputs move by their rounding resolution: agent means
        and node probabilities by +-1 float32 ulp, the ego state by a relative 1e-13 (two draws each)"""
        cw, cf, flats, x0, lane, tv = got["args"]
        ref = np.stack([base[:, 0], base[:, 1], np.where(base[:, 2] >= 0, 1.0, base[:, 2])], axis=1)
        best = len(ref)
        for seed in range(6):
            rng = np.random.default_rng(seed)
            f2, x2 = dict(flats[ti]), np.array(x0, dtype=np.float64)
            if seed < 4:
                key = "mean" if seed < 2 else "prob"
                m = np.asarray(flats[ti][key], np.float32)
                f2[key] = np.where(rng.random(m.shape) < 0.5, np.nextafter(m, np.float32(np.inf)), np.nextafter(m, np.float32(-np.inf))).astype(np.float32)
            else:
                x2 = x2 * (1.0 + 1e-13 * rng.standard_normal(x2.shape))
            rt.ilqr_contingency(cw, cf, [f2], x2, lane, tv)
            k2 = _first_divergence(rt.ilqr_trace(0, ph), ref)
            best = min(best, len(ref) if k2 is None else k2)
        return best

    def oracle_view(ti, ph, dev_trace, ref):
        """The same fit through the C oracle, the float64 restatement on the CPU (same operations in the same order, same sin / cos /
        tan routine: mind_amd/csrc/mind_trig.h): (the oracle follows the reference's trace further than the device does, the oracle's
        trace equals the device's iteration by iteration).  An `own_noise` fit on which the oracle gives the device's trace is the cost
        tree's own sensitivity to rounding noise in its inputs, not something the kernel adds."""
        from oracle import ilqr as oi
        cw, cf, flats, x0, lane, tv = got["args"]
        wsol = oi.solve(cw, flats[ti], x0, lane, tv, 0, trace=True)
        sol = wsol if ph == 0 else oi.solve(cf, flats[ti], x0, lane, tv, 1, us_init=wsol["us"], trace=True)
        ko, kd = _first_divergence(sol["trace"], ref), _first_divergence(dev_trace, ref)
        far = lambda k: 10 ** 9 if k is None else k
        same_as_dev = len(sol["trace"]) == len(dev_trace) and np.array_equal(sol["trace"][:, [0, 2]], dev_trace[:, [0, 2]]) \
            and np.allclose(sol["trace"][:, 1], dev_trace[:, 1], rtol=1e-12, atol=0)
        return far(ko) > far(kd), same_as_dev

    opt.solve_batch = capture
    state_in, ctrl_in = D[scene + "_state_in"], D[scene + "_ctrl_in"]
    n_fits = same = behind_split = behind_warm = own_noise = iters = 0
    early, noise_cause = [], []
    for pi in range(60):
        while sim.n_plans <= pi:
            will_plan = sim.sim_time >= sim.enable_time and (sim.last_trigger is None or sim.sim_time - sim.last_trigger >= sim.PLAN_STEP)
            if will_plan and sim.enabled and (branching or pi > 0):
                sim.state, sim.ctrl = state_in[pi].copy(), ctrl_in[pi].copy()
            sim.step()
        n_trees = len(got["traces"])
        assert (pi, n_trees - 1, 1) in fits and (pi, n_trees, 0) not in fits, (pi, n_trees)      # same number of candidates as the reference
        for ti, tr in enumerate(got["traces"]):
            warm_left = False
            for ph in (0, 1):
                ref, sp = fits[(pi, ti, ph)]
                n_fits += 1
                iters += len(ref)
                k = _first_divergence(tr[ph], ref)
                if k is None:
                    same += 1
                elif warm_left:
                    behind_warm += 1
                elif sp < len(ref) and k >= sp - 2:
                    behind_split += 1
                elif own_split(ti, ph, tr[ph]) <= k + 2:
                    own_noise += 1          # the two perturbed reference runs of the golden did not part that early, this solver's own do
                    noise_cause.append((pi, ti, ph, k) + oracle_view(ti, ph, tr[ph], ref))
                else:
                    early.append((pi, ti, ph, k, sp, len(ref), len(tr[ph])))
                if ph == 0 and k is not None:
                    warm_left = True
    print(f"[{scene} {variant}] {n_fits} fits / {iters} reference iterations: {same} identical traces, {behind_split} part where the reference's own "
          f"perturbed runs part, {own_noise} where this solver's own perturbed runs part, {behind_warm} full fits behind such a warm start, "
          f"unexplained: {early}")
    if noise_cause:
        print(f"[{scene} {variant}] the {own_noise} own-noise fits through the C oracle: the oracle's trace equals the device's in "
              f"{sum(1 for c in noise_cause if c[5])}, the oracle follows the reference further than the device in {sum(1 for c in noise_cause if c[4])}; "
              f"(cycle, tree, phase, first parting iteration, oracle further, oracle == device): {noise_cause}")
    assert not early, early
    # every fit on which the reference reproduces ITSELF under rounding noise is followed iteration by iteration, up to the few that only
    # this solver's own noise runs explain (the golden holds two perturbed reference runs per fit; observed: <= 2.3 % of a scene's fits)
    n_ref_split = int(sum(1 for (r_, sp_) in fits.values() if sp_ < len(r_)))
    # own-noise fits: which fits these are depends on the last bits of the predictor's output and of sin / cos / tan (three builds: 13, 18
    # and 18 of 2 178 over the eight (scene, weights) runs, profiles/r04j_*, r04w_*, r04aj_*; the worst run has 10 of 432 = 2.3 %).  Since
    # kernel and oracle share their trigonometric routine the C oracle gives the device's trace on every one of them: the cost trees' own
    # sensitivity to rounding noise in their inputs (with the device library's functions in the kernel and glibc's in the oracle, 2 of 18
    # were fits only the oracle followed)
    assert same + own_noise + behind_warm >= n_fits - n_ref_split and own_noise <= max(2, 0.025 * n_fits), (same, own_noise, n_ref_split, n_fits)
    assert all(c[5] for c in noise_cause), noise_cause
