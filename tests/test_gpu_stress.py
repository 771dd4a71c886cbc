"""BASELINE configs[4] as a workload: 128 agents x 256 lane polylines (N = 385) on the largest tree the reference's probability
floor lets grow (full scripted 6-ary depth-4 AIME tree on the real predictor forward, 259 expansions per plan: `stress128tree` of
bench.py), end to end -- predictor, AIME, tree-iLQR with several workgroups per cost tree -- in the default arithmetic and in the
plain bf16 mode the config names.  No CPU oracle finishes at this size in test time, so the checks are the size-independent
properties: expansion counts, sibling probabilities summing to one, finite world-frame rows, converged / improving solves, equal
tree structure between the two arithmetics and a bounded difference of the ego plans."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _plans(prec, n=2):
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    pl, sim, w = make_closed_loop(dict(WORKLOADS["stress128tree"]), full_tree=True, speculative=False)
    rt = pl.network.rt
    before = rt.pair_precision()
    out = []
    try:
        rt.set_pair_precision(prec)
        for _ in range(n):
            e0 = pl.scen_tree_gen.n_expanded
            sim.run_plans(1)
            trees = pl.scen_tree_gen.get_scenario_tree()
            scen, traj = sim.last_result
            out.append(dict(expanded=pl.scen_tree_gen.n_expanded - e0, trees=trees, best=pl.timing["best_traj_idx"], costs=pl.timing["tree_costs"],
                            xs=np.array([n_.data[0] for k, n_ in traj[0].nodes.items() if k != -1]), ctrl=np.array(sim.ctrl),
                            debug=pl.traj_tree_opt.debug, wgs=rt.ilqr_stats()[2]))
    finally:
        rt.set_pair_precision(before)
    return out


def test_stress_workload_end_to_end_in_both_arithmetics():
    a = _plans("bf16x3")
    b = _plans("bf16")
    for runs in (a, b):
        for p in runs:
            assert p["expanded"] == 259                                           # 1 + 6 + 36 + 216 scenes through the predictor
            assert len(p["trees"]) == 6 and p["wgs"] >= 2                           # six scenario trees, several workgroups per cost tree
            assert 100 <= sum(len(t.nodes) for t in p["trees"]) <= 1555             # nodes of finished branches (at most 6 + 36 + 216 + 1296)
            for t in p["trees"]:
                for k, n in t.nodes.items():
                    pos, cov = np.asarray(n.data[1]), np.asarray(n.data[2])
                    assert pos.shape[0] == 128 and pos.shape[2] == 2 and np.isfinite(pos).all() and np.isfinite(cov).all() and (cov > 0).all()
                    ch = [t.nodes[c] for c in n.children_keys]
                    if ch:                                                         # sibling probabilities: renormalised to the parent's
                        assert abs(sum(float(np.ravel(c.data[0])[0]) for c in ch) - float(np.ravel(n.data[0])[0])) < 1e-5
            assert np.isfinite(p["xs"]).all() and np.isfinite(p["costs"]).all() and len(p["costs"]) == 6
            assert all(s["iterations"] >= 1 for s in p["debug"]["full"]) and all(np.isfinite(s["J"]) for s in p["debug"]["full"])
    # same tree structure in both arithmetics (the scripted modes fix the branching), ego plans close
    for pa, pb in zip(a, b):
        assert [list(t.nodes.keys()) for t in pa["trees"]] == [list(t.nodes.keys()) for t in pb["trees"]]
    d = float(np.abs(a[0]["xs"][:, :2] - b[0]["xs"][:, :2]).max()) if a[0]["best"] == b[0]["best"] and a[0]["xs"].shape == b[0]["xs"].shape else float("nan")
    print(f"stress128tree, first plan: max |ego xy (bf16x3) - ego xy (bf16)| = {d:.3e} m, chosen trees {a[0]['best']} / {b[0]['best']}")
    assert not (d > 5.0)


def test_chunked_rounds_plan_what_whole_rounds_plan():
    """mind_aime_plan sends a round whose edge tensor would exceed the budget (mind_set_tuning "plan_chunk_mb") through the predictor in
    chunks of scenes (the scenes of a round are independent: network.py:318,497).  The full cfg4 tree with a 2 GB budget (the 216-scene
    round in six chunks of 37 scenes, the 36-scene round whole) must plan bit for bit what the unchunked plan plans."""
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    outs = []
    for mb in (96 * 1024, 2048):
        pl, sim, w = make_closed_loop(dict(WORKLOADS["cfg4tree"]), full_tree=True, speculative=False)
        rt = pl.network.rt
        try:
            rt.set_tuning("plan_chunk_mb", mb)
            e0 = pl.scen_tree_gen.n_expanded
            sim.run_plans(1)
        finally:
            rt.set_tuning("plan_chunk_mb", 96 * 1024)
        scen, traj = sim.last_result
        trees = pl.scen_tree_gen.get_scenario_tree()
        outs.append(dict(expanded=pl.scen_tree_gen.n_expanded - e0, native=pl.scen_tree_gen.n_native_plans, keys=[list(t.nodes.keys()) for t in trees],
                         pos=[np.asarray(n.data[1]) for t in trees for n in t.nodes.values()], xs=np.array([n_.data[0] for k, n_ in traj[0].nodes.items() if k != -1]),
                         costs=np.array(pl.timing["tree_costs"])))
    a, b = outs
    assert a["expanded"] == b["expanded"] == 259 and a["native"] >= 1 and b["native"] >= 1 and a["keys"] == b["keys"]
    assert all(np.array_equal(x, y) for x, y in zip(a["pos"], b["pos"])) and np.array_equal(a["xs"], b["xs"]) and np.array_equal(a["costs"], b["costs"])


def test_deep_tree_stress_workload_one_plan():
    """`stressdeep`: 128 agents x 256 lane polylines under the scripted 6-ary depth-5 tree with the probability floor lifted (rounds of 1 / 6 /
    36 / 216 / 1 296 scenes = 1 555 expansions, 7 776 leaves -- what K = 6 modes and max_depth = 5 leave of BASELINE configs[4]'s 8-ary depth-6
    tree), in the plain bf16 arithmetic the config names, the 1 296-scene round in chunks under a 24 GB edge budget: the size-independent properties
    of one plan (no oracle finishes here)."""
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    pl, sim, w = make_closed_loop(dict(WORKLOADS["stressdeep"]), full_tree="deep", speculative=False)
    rt = pl.network.rt
    before = rt.pair_precision()
    try:
        rt.set_pair_precision("bf16")
        rt.set_tuning("plan_chunk_mb", 24 * 1024)
        e0 = pl.scen_tree_gen.n_expanded
        sim.run_plans(1)
    finally:
        rt.set_pair_precision(before)
        rt.set_tuning("plan_chunk_mb", 96 * 1024)
    info = rt.last_aime_info
    assert pl.scen_tree_gen.n_expanded - e0 == 1555 and info["round_scenes"] == [1, 6, 36, 216, 1296]
    trees = pl.scen_tree_gen.get_scenario_tree()
    assert len(trees) == 6 and sum(len(t.nodes) for t in trees) == 6 + 36 + 216 + 1296 + 7776
    for t in trees:
        for k, n in t.nodes.items():
            pos, cov = np.asarray(n.data[1]), np.asarray(n.data[2])
            assert pos.shape[0] == 128 and np.isfinite(pos).all() and np.isfinite(cov).all() and (cov > 0).all()
            ch = [t.nodes[c] for c in n.children_keys]
            if ch:
                assert abs(sum(float(np.ravel(c.data[0])[0]) for c in ch) - float(np.ravel(n.data[0])[0])) < 1e-5
    scen, traj = sim.last_result
    xs = np.array([n_.data[0] for k, n_ in traj[0].nodes.items() if k != -1])
    assert np.isfinite(xs).all() and len(pl.timing["tree_costs"]) == 6 and np.isfinite(pl.timing["tree_costs"]).all()
    assert all(s["iterations"] >= 1 and np.isfinite(s["J"]) for s in pl.traj_tree_opt.debug["full"])
    print(f"stressdeep: 1 555 expansions, {sum(len(t.nodes) for t in trees)} scenario-tree nodes, cost trees of {[len(t.nodes) for t in traj[:1]]} .. trajectory nodes; "
          f"aime {pl.timing['aime_s'] * 1e3:.0f} ms, tree-iLQR {pl.timing['ilqr_s'] * 1e3:.0f} ms")


def test_deeper_tree_stress_workload_one_plan():
    """`stressdeeper`: the same scene under the scripted 6-ary depth-6 tree -- six AIME rounds of 1 / 6 / 36 / 216 / 1 296 / 7 776 scenes = 9 331
    expansions, 46 656 leaves: the depth BASELINE configs[4] names at the branching K = 6 modes allow.  The last round's edge tensor alone
    (7 776 scenes x 39 MB in plain bf16) exceeds the 288 GB of the device: it runs in chunks under a 64 GB budget with about 200 GB in use.
    Size-independent properties of the one plan: round sizes, node counts, probabilities of siblings summing to their parent's, finite
    trajectories / covariances / costs."""
    import torch
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    pl, sim, w = make_closed_loop(dict(WORKLOADS["stressdeeper"]), full_tree="deeper", speculative=False)
    rt = pl.network.rt
    before = rt.pair_precision()
    try:
        rt.set_pair_precision("bf16")
        rt.set_tuning("plan_chunk_mb", 64 * 1024)
        e0 = pl.scen_tree_gen.n_expanded
        sim.run_plans(1)
    finally:
        rt.set_pair_precision(before)
        rt.set_tuning("plan_chunk_mb", 96 * 1024)
    info = rt.last_aime_info
    assert pl.scen_tree_gen.n_expanded - e0 == 9331 and info["round_scenes"] == [1, 6, 36, 216, 1296, 7776]
    trees = pl.scen_tree_gen.get_scenario_tree()
    assert len(trees) == 6 and sum(len(t.nodes) for t in trees) == 6 + 36 + 216 + 1296 + 7776 + 46656
    t = trees[0]
    for k, n in t.nodes.items():
        ch = [t.nodes[c] for c in n.children_keys]
        if ch:
            assert abs(sum(float(np.ravel(c.data[0])[0]) for c in ch) - float(np.ravel(n.data[0])[0])) < 1e-5
    leaves = [n for n in t.nodes.values() if not n.children_keys][:200]
    for n in leaves:
        pos, cov = np.asarray(n.data[1]), np.asarray(n.data[2])
        assert pos.shape[0] == 128 and np.isfinite(pos).all() and np.isfinite(cov).all() and (cov > 0).all()
    scen, traj = sim.last_result
    xs = np.array([n_.data[0] for k, n_ in traj[0].nodes.items() if k != -1])
    assert np.isfinite(xs).all() and len(pl.timing["tree_costs"]) == 6 and np.isfinite(pl.timing["tree_costs"]).all()
    free, tot = torch.cuda.mem_get_info()
    print(f"stressdeeper: 9 331 expansions, {sum(len(t_.nodes) for t_ in trees)} scenario-tree nodes, cost trees of {len(traj[0].nodes)} trajectory nodes; "
          f"aime {pl.timing['aime_s']:.2f} s (first plan: arenas grow), {(tot - free) / 2 ** 30:.0f} GiB of device memory in use")
