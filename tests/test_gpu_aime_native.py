"""mind_aime_plan (the whole AIME loop of ScenarioTreeGenerator.branch_aime, scenario_tree.py:38-58, in one native call: per-round
bookkeeping in C++, branch-time test / windows / re-basing on the device) against the round-by-round host path over the same
kernels: the internal tree (node ids in insertion order, CUR_T / END_T, branch / end / terminate flags) and every returned scenario
tree (sibling-normalised probabilities, agent trajectories, covariances, target windows) must be bit-identical, cycle by cycle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _flat(trees):
    out = []
    for t in trees:
        for k, n in t.nodes.items():
            out.append((k, n.parent_key, np.asarray(n.data[0]).copy(), np.asarray(n.data[1]).copy(), np.asarray(n.data[2]).copy(),
                        np.asarray(n.data[3]).copy()))
    return out


@pytest.mark.parametrize("scene", ["demo_1", "demo_3", "demo1", "small_full_tree"])
def test_native_plan_equals_the_round_by_round_path(scene):
    """demo_1 / demo_3: recorded scenes, the predictor's own modes (branching formula weights).  demo1 / small_full_tree: synthetic
    worlds with SCRIPTED modes on top of the real forward (mind_amd/synth.py; mind_aime_plan_in.script_*: the bench's demo-like
    branching and the full 6-ary depth-4 tree of BASELINE config 4 -- 259 expansions, rounds of 1 / 6 / 36 / 216 scenes -- on a
    16-agent world)."""
    sys.path.insert(0, ROOT)
    from bench import BRANCHING_WEIGHTS, WORKLOADS, make_closed_loop
    sims = []
    synthetic = scene in ("demo1", "small_full_tree")
    wkw = dict(n_agents=16, n_lanes=4, n_segs=8, seed=4) if scene == "small_full_tree" else dict(WORKLOADS[scene])
    for native in (True, False):
        pl, sim, w = make_closed_loop(wkw, ckpt=None if synthetic else BRANCHING_WEIGHTS, speculative=False, full_tree=scene == "small_full_tree")
        pl.scen_tree_gen.native_aime = native
        pl.scen_tree_gen.device_root = False         # both fed by the host featuriser (the device-built root is compared below)
        sims.append((pl, sim))
    n_multi = 0
    n_cycles = 3 if scene == "small_full_tree" else 8
    for cycle in range(n_cycles):
        res = []
        for pl, sim in sims:
            sim.run_plans(1)
            gen = pl.scen_tree_gen
            internal = [(k, n.parent_key, int(n.data.data["CUR_T"]) if n.data.data is not None else -1,
                         int(n.data.data["END_T"]) if n.data.data is not None else -1,
                         bool(n.data.branch_flag), bool(n.data.end_flag), bool(n.data.terminate_flag)) for k, n in gen.tree.nodes.items()]
            res.append((internal, _flat(gen.get_scenario_tree()), np.array(sim.ctrl), pl.timing["nodes_expanded"], pl.timing["best_traj_idx"]))
            if gen.native_aime:
                # the cost trees the library flattened (trajectory_tree.py:19-124) = what flatten_scenario_tree builds from the returned trees
                from mind_amd.planners.mind.trajectory_tree import flatten_scenario_tree
                for t in gen.last_trees:
                    f, g_ = flatten_scenario_tree(t), t._flat
                    assert g_ is not None and all(np.array_equal(f[k], g_[k]) and f[k].dtype == g_[k].dtype for k in ("parent", "prob", "mean", "cov")), cycle
        (ia, fa, ca, ea, ba), (ib, fb, cb, eb, bb) = res
        assert ia == ib, (cycle, ia, ib)
        assert len(fa) == len(fb) and ea == eb and ba == bb
        for x, y in zip(fa, fb):
            assert x[0] == y[0] and x[1] == y[1], (cycle, x[0], y[0])
            for u, v in zip(x[2:], y[2:]):
                assert u.dtype == v.dtype and u.shape == v.shape and np.array_equal(u, v), (cycle, x[0])
        assert np.array_equal(ca, cb)
        n_multi += len(ia) > 3
    assert sims[0][0].scen_tree_gen.n_native_plans == n_cycles and sims[1][0].scen_tree_gen.n_native_plans == 0
    assert n_multi >= n_cycles // 2     # the branching weights / scripted modes really grow multi-round trees here
    if scene == "small_full_tree":
        assert sims[0][0].timing["nodes_expanded"] >= 250          # 1 + 6 + 36 + 216 when no mode is pruned


@pytest.mark.parametrize("scene", ["demo_1", "demo_2", "demo_4"])
def test_device_built_root_equals_the_host_featuriser(scene):
    """process_data / prepare_root_data as kernels (k_aime_rebase on the raw windows, k_aime_root_lanes, k_aime_root_hist) against
    the host featuriser feeding the same native plan: the same float32 / float64 expressions, so every discrete result of the
    AIME tree (node ids, CUR_T / END_T, flags) must be equal and the arrays agree to float32 rounding of the ~6 km map coordinates; the chosen
    tree is equal too except where a candidate's tree-iLQR fit is ill-conditioned under that rounding noise (probed, at most one cycle in ten)."""
    sys.path.insert(0, ROOT)
    from bench import BRANCHING_WEIGHTS, WORKLOADS, make_closed_loop
    from mind_amd.planners.mind.trajectory_tree import flatten_scenario_tree, ilqr_cfg_from
    from test_gpu_plan import _solution_moves_under_rounding_noise
    sims, caps = [], []
    for dev_root in (True, False):
        pl, sim, w = make_closed_loop(dict(WORKLOADS[scene]), ckpt=BRANCHING_WEIGHTS, speculative=False)
        pl.scen_tree_gen.device_root = dev_root
        sims.append((pl, sim))
        cap, opt = {}, pl.traj_tree_opt

        def capture(scen_trees, init_state, init_ctrl, target_lane, target_vel, cap=cap, opt=opt, orig=opt.solve_batch):
            trees = orig(scen_trees, init_state, init_ctrl, target_lane, target_vel)     # what the contingency solves were given
            cap["common"] = (ilqr_cfg_from(opt.config, "w_opt_cfg"), ilqr_cfg_from(opt.config, "opt_cfg"),
                             opt._get_init_state(init_state, init_ctrl), np.asarray(target_lane, np.float64), target_vel)
            cap["scen_trees"], cap["xs"] = scen_trees, [t._arrays[0][1:] for t in trees]
            return trees
        opt.solve_batch = capture
        caps.append(cap)
    ulp = float(np.spacing(np.float32(np.abs(sims[0][1].world.pos[0, 0]).max())))
    worst, ill = 0.0, []
    for cycle in range(10):
        res = []
        for pl, sim in sims:
            sim.run_plans(1)
            gen = pl.scen_tree_gen
            internal = [(k, n.parent_key, int(n.data.data["CUR_T"]) if n.data.data is not None else -1,
                         int(n.data.data["END_T"]) if n.data.data is not None else -1,
                         bool(n.data.branch_flag), bool(n.data.end_flag), bool(n.data.terminate_flag)) for k, n in gen.tree.nodes.items()]
            res.append((internal, _flat(gen.get_scenario_tree()), np.array(sim.ctrl), pl.timing["best_traj_idx"]))
        (ia, fa, ca, ba), (ib, fb, cb, bb) = res
        assert ia == ib, (cycle, ia, ib)
        assert len(fa) == len(fb)
        if ba != bb:
            # the two roots differ by float32 rounding of the inputs: another chosen tree is admissible only where the fit of the candidate
            # the two runs price differently is ill-conditioned under exactly that noise (the probe of tests/test_gpu_plan.py)
            ka, kb = np.array(sims[0][0].timing["tree_costs"]), np.array(sims[1][0].timing["tree_costs"])
            j = int(np.argmax(np.abs(ka - kb)))
            cw, cf, x0, lane, tv = caps[0]["common"]
            moved = _solution_moves_under_rounding_noise(sims[0][0].network.rt.ilqr_contingency,
                                                         (cw, cf, [flatten_scenario_tree(caps[0]["scen_trees"][j])], x0, lane, tv), caps[0]["xs"][j])
            print(f"{scene} cycle {cycle}: chosen tree {ba} (device root) / {bb} (host root); costs {ka} / {kb}; candidate {j}'s own solution "
                  f"moves by {moved:.2e} m under rounding noise")
            assert moved > 1e-3 + 2 * ulp, (cycle, ba, bb, ka, kb, moved)
            ill.append(cycle)
        for x, y in zip(fa, fb):
            assert x[0] == y[0] and x[1] == y[1]
            assert abs(float(np.ravel(x[2])[0]) - float(np.ravel(y[2])[0])) < 1e-4
            worst = max(worst, float(np.abs(x[3] - y[3]).max()))
            assert np.abs(x[3] - y[3]).max() < 2e-3 + 4 * ulp and np.abs(x[4] - y[4]).max() < 1e-3 and np.array_equal(x[5], y[5]), (cycle, x[0])
        sims[1][1].state, sims[1][1].ctrl = sims[0][1].state.copy(), np.array(sims[0][1].ctrl).copy()      # keep the two loops on one trajectory
    print(f"{scene}: device-built vs host-built root, max |agent position difference| over 10 cycles = {worst:.2e} m (float32 ulp there {ulp:.1e})")
    assert sims[0][0].scen_tree_gen.n_native_plans == 10 and len(ill) <= 1, ill


def test_lazy_scenario_tree_read_after_the_next_plan_is_still_its_own_plans():
    """The drop-in path stores frame['scen_tree'] = res[0] and reads it at render time (the reference's simulator.py:92-107): the LazyTree a plan
    returns must materialise from THAT plan's AIME tree even when it is first touched after the generator has planned (or been reset) again."""
    sys.path.insert(0, ROOT)
    from bench import BRANCHING_WEIGHTS, WORKLOADS, make_closed_loop
    runs = []
    for late in (False, True):
        pl, sim, w = make_closed_loop(dict(WORKLOADS["demo_1"]), ckpt=BRANCHING_WEIGHTS, speculative=False)
        kept = []
        for cycle in range(4):
            sim.run_plans(1)
            trees = pl.scen_tree_gen.last_trees
            assert trees is not None
            if not late:
                kept.append(_flat(trees))                 # materialised at once
            else:
                kept.append(trees)                        # ... or only after every later plan and a reset
        if late:
            pl.scen_tree_gen.reset()
            kept = [_flat(t) for t in kept]
        runs.append(kept)
    for a, b in zip(*runs):
        assert len(a) == len(b) > 0
        for x, y in zip(a, b):
            assert x[0] == y[0] and x[1] == y[1] and all(np.array_equal(p, q) for p, q in zip(x[2:], y[2:]))
