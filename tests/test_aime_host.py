"""CPU: the AIME scenario-tree bookkeeping (mind_amd.planners.mind.scenario_tree) driven by the scripted
FakeNet, against golden decisions captured from the reference ScenarioTreeGenerator (node-id sets,
parents, sibling probabilities, window lengths, world-frame trajectories)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fake_net import FakeNet  # noqa: E402

from mind_amd.planners.basic.tree import Node, Tree
from mind_amd.planners.mind import utils as U
from mind_amd.planners.mind.configs.planning.demo_1 import ScenTreeCfg
from mind_amd.planners.mind.planner import MINDPlanner
from mind_amd.planners.mind.scenario_tree import ScenarioTreeGenerator
from mind_amd.synth import SynthWorld

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = dict(np.load(os.path.join(ROOT, "tests", "golden", "aime.npz")))
CASES = [
    ("w6", dict(n_agents=6, n_lanes=3, n_segs=8, seed=1), dict()),
    ("w3_nogrowth", dict(n_agents=3, n_lanes=2, n_segs=6, seed=2), dict(growth=(0.02,) * 6)),
    ("w9_branch", dict(n_agents=9, n_lanes=3, n_segs=8, seed=3),
     dict(lateral=(0.0, 9.0, -9.0, 0.1, -6.0, 0.2), growth=(0.3, 0.25, 0.1, 0.02, 0.4, 0.03))),
    ("w5_deep", dict(n_agents=5, n_lanes=3, n_segs=8, seed=4),
     dict(lateral=(0.0, 9.0, -9.0, 4.0, -6.0, 0.2), growth=(0.5, 0.45, 0.4, 0.5, 0.4, 0.3),
          probs=(0.3, 0.25, 0.2, 0.15, 0.0995, 0.0005))),
    ("w1_ego_only", dict(n_agents=1, n_lanes=2, n_segs=6, seed=5), dict()),
]


def run_aime(wkw, nkw):
    w = SynthWorld(**wkw)
    lcl = w.local_semantic_map(4.9)
    obs = w.tracks(4.9, drop={2: 30} if wkw["n_agents"] > 2 else None)
    lane, info = MINDPlanner.resample_target_lane(MINDPlanner.__new__(MINDPlanner), lcl)
    net = FakeNet(**nkw)
    g = ScenarioTreeGenerator(torch.device("cpu"), net, 50, 50, ScenTreeCfg())
    g.reset()
    g.set_target_lane(lane, info)
    return g, net, g.branch_aime(lcl, obs)


@pytest.mark.parametrize("name,wkw,nkw", CASES)
def test_aime_matches_reference_golden(name, wkw, nkw):
    g, net, trees = run_aime(wkw, nkw)
    assert list(net.calls) == list(G[name + "_batches"])                         # scenes per AIME round
    assert list(g.tree.nodes.keys()) == list(G[name + "_internal_keys"])         # identical node-id set/order
    flags = np.array([[n.data.branch_flag, n.data.end_flag, n.data.terminate_flag] for n in g.tree.nodes.values()])
    assert np.array_equal(flags, G[name + "_internal_flags"])
    assert len(trees) == int(G[name + "_ntrees"])
    for ti, t in enumerate(trees):
        keys = list(t.nodes.keys())
        assert keys == list(G[f"{name}_t{ti}_keys"])
        assert [str(t.nodes[k].parent_key) for k in keys] == list(G[f"{name}_t{ti}_parents"])
        assert [t.nodes[k].data[1].shape[1] for k in keys] == list(G[f"{name}_t{ti}_durs"])      # END_T - CUR_T
        probs = np.array([float(np.ravel(t.nodes[k].data[0])[0]) for k in keys])
        assert np.abs(probs - G[f"{name}_t{ti}_probs"]).max() < 1e-6
        for k in keys:
            d = t.nodes[k].data
            assert d[1].dtype == np.float32 and d[2].dtype == np.float32
            assert np.abs(d[1][:, ::5] - G[f"{name}_t{ti}_{k}_pos"]).max() < 1e-3       # metres, world frame
            assert np.abs(d[2][:, ::5] - G[f"{name}_t{ti}_{k}_cov"]).max() < 1e-5
            assert np.abs(np.asarray(d[3]) - G[f"{name}_t{ti}_{k}_tgt"]).max() < 1e-4


def test_tracks_padding_and_order():
    w = SynthWorld(n_agents=4, n_lanes=2, n_segs=6, seed=1)
    obs = w.tracks(4.9, drop={2: 30})
    # make agent 3 unobserved at the last step: it must be skipped
    last = obs[w.agent_ids[3]].object_states[-1]
    last.observed = False
    pos, ang, vel, typ, flags, tids, cats = U.get_agent_trajectories(obs)
    assert tids == ["AV", w.agent_ids[1], w.agent_ids[2]] and cats == ["av", "exo", "exo"]
    assert pos.shape == (3, 50, 2) and pos.dtype == np.float32 and flags.dtype == np.int16
    assert flags[2].sum() == 20 and np.all(flags[2][:30] == 0)
    assert np.all(pos[2, :30] == pos[2, 30]) and np.all(vel[2, :30] == 0)       # NN pad pos, zero pad vel
    assert np.all(typ[2, :30] == 0) and typ[2, 30:].sum() == 20


def test_lane_graph_shapes_and_frames():
    w = SynthWorld(n_agents=2, n_lanes=3, n_segs=5, seg_len=32.0, seed=1)
    g = U.lane_graph_from_map(w, np.array([10.0, 1.0], np.float32), U.rot2(np.float32(0.1)))
    assert g["num_lanes"] == 3 * 5 * 2          # 32 m segments -> floor(32.x/15) = 2 pieces each
    assert g["node_ctrs"].shape == (30, 10, 2) and g["node_ctrs"].dtype == np.float32
    assert np.abs(g["node_ctrs"].mean(axis=1)).max() < 1e-2            # instance frame is centred
    assert np.allclose(np.linalg.norm(g["lane_vecs"], axis=1), 1.0, atol=1e-6)
    f = U.lane_features(g)
    assert f.shape == (30, 10, 16) and set(np.unique(f[..., 4:])) <= {0.0, 1.0}


def test_tree_container_contract():
    t = Tree()
    t.add_node(Node("r", None, 0))
    t.add_node(Node("a", "r", 1))
    t.add_node(Node("b", "r", 2))
    t.add_node(Node("c", "a", 3))
    assert t.leaves == ["b", "c"] and t.get_node("c").depth == 2 and t.size() == 4
    assert [n.key for n in t.retrieve_nodes_to_root("c")] == ["c", "a", "r"]
    with pytest.raises(KeyError):
        t.add_node(Node("x", "nope", 0))
    with pytest.raises(ValueError):
        t.add_node(Node("a", "r", 0))
    with pytest.raises(KeyError):
        t.get_node("zz")


def test_agent_trajectories_vectorised_path_equals_per_agent_loop(monkeypatch):
    """full-window fast path of get_agent_trajectories (all agents at once) == the per-agent loop, incl. unobserved gaps."""
    from types import SimpleNamespace
    from mind_amd.planners.mind import utils as U
    rng = np.random.default_rng(0)
    agent_obs = {}
    for i, key in enumerate(["x3", "AV", "x1", "x2", "x7", "s9", "s1"]):
        states, rows = [], []
        t_first = {"s9": 41, "s1": 49}.get(key, 0)             # short tracks: agents that appeared 0.9 s / just now
        for t in range(t_first, 50):
            seen = not (key == "x1" and t < 7) and not (key == "x2" and 20 <= t < 26) and not (key == "x7" and t == 49)
            p, h, v = rng.normal(size=2) * 30, rng.uniform(-3, 3), rng.normal(size=2) * 4
            states.append(SimpleNamespace(observed=seen, timestep=t, position=(p[0], p[1]), heading=h, velocity=(v[0], v[1])))
            rows.append([float(seen), p[0], p[1], h, v[0], v[1]])
        agent_obs[key] = SimpleNamespace(track_id=key, object_states=states, object_type=["vehicle", "pedestrian", "bus"][i % 3],
                                         category=None, _arr=np.array(rows))
    fast = U.get_agent_trajectories(agent_obs)
    monkeypatch.setattr(U, "_agent_trajectories_full_windows", lambda *a: None)
    slow = U.get_agent_trajectories(agent_obs)
    assert fast[5] == slow[5] == ["AV", "x3", "x1", "x2", "s9", "s1"] and fast[6] == slow[6]   # x7 unobserved now: skipped; AV first
    assert fast[4][4].sum() == 9 and fast[4][5].sum() == 1                                      # short tracks sit at the end of the window
    for f, s_ in zip(fast[:5], slow[:5]):
        assert f.dtype == s_.dtype and np.array_equal(f, s_)


def test_distances_to_polyline_batched_equals_single():
    from mind_amd.planners.mind import utils as U
    rng = np.random.default_rng(1)
    lane = np.cumsum(rng.uniform(0.5, 2.0, (30, 2)), axis=0).astype(np.float32)
    pts = (rng.uniform(0, 40, (6, 2))).astype(np.float32)
    d = U.get_distances_to_polyline(lane, pts)
    assert d.dtype == np.float32
    for i in range(6):
        assert d[i] == U.get_distance_to_polyline(lane, pts[i])


def _full_tree_run(batched):
    """CPU run of the scripted 6-ary tree (16 agents) through the AIME bookkeeping with a stub network."""
    from mind_amd.synth import ScriptedFullTree

    class Stub:
        computes_rpe_in_kernel = True
        last_packed = None
        last_lane_feat = None

    class CpuFull(ScriptedFullTree):
        def pre_process(self, d):
            return d

        def __call__(self, d):
            cls, reg, aux = [], [], []
            for ii in d["ACTOR_IDCS"]:
                c, r, v = self._modes(len(ii), "cpu")
                cls.append(c); reg.append(r); aux.append((v, None, None))
            return cls, reg, aux

    w = SynthWorld(n_agents=16, n_lanes=8, n_segs=32, seed=4)
    lcl = w.local_semantic_map(4.9)
    obs = w.tracks(4.9)
    lane, info = MINDPlanner.resample_target_lane(MINDPlanner.__new__(MINDPlanner), lcl)
    g = ScenarioTreeGenerator(torch.device("cpu"), CpuFull(Stub()), 50, 50, ScenTreeCfg())
    if not batched:
        g.update_obser_batch = lambda curs, own=None: [g.update_obser(c) for c in curs]
        g.get_branch_times = lambda datas: [g.get_branch_time(d) for d in datas]
    g.reset()
    g.set_target_lane(lane, info)
    trees = g.branch_aime(lcl, obs)
    return g, trees


def test_batched_round_bookkeeping_equals_per_node_path():
    """decide_branch's batched branch-time search and observation re-basing (all branching nodes of a round in one
    set of array ops) give bit-identical node data and scene inputs to the per-node functions, on the scripted
    6-ary depth-4 tree (253 expansions)."""
    gb, tb = _full_tree_run(True)
    gl, tl = _full_tree_run(False)
    assert gb.n_expanded == gl.n_expanded > 200
    assert list(gb.tree.nodes.keys()) == list(gl.tree.nodes.keys())
    for k in gb.tree.nodes:
        db, dl = gb.tree.nodes[k].data, gl.tree.nodes[k].data
        assert (db.branch_flag, db.end_flag, db.terminate_flag) == (dl.branch_flag, dl.end_flag, dl.terminate_flag)
        for part_b, part_l in ((db.data, dl.data), (db.obs_data, dl.obs_data)):
            assert (part_b is None) == (part_l is None)
            if part_b is None:
                continue
            assert set(part_b.keys()) == set(part_l.keys())
            for f in part_b:
                vb, vl = part_b[f], part_l[f]
                if isinstance(vb, np.ndarray):
                    assert vb.dtype == vl.dtype and vb.shape == vl.shape and np.array_equal(vb, vl), (k, f)
                elif not isinstance(vb, (list, str, type(None))):
                    assert vb == vl, (k, f)
    assert [len(t.nodes) for t in tb] == [len(t.nodes) for t in tl]


def test_batched_tree_evaluation_equals_per_tree():
    """evaluate_traj_trees (all candidate trees in one pass) == evaluate_traj_tree per tree (planner.py:180-198)."""
    from types import SimpleNamespace
    from mind_amd.planners.mind.trajectory_tree import to_traj_tree
    rng = np.random.default_rng(2)
    pl = MINDPlanner.__new__(MINDPlanner)
    lane = np.cumsum(rng.uniform(0.5, 2.0, (40, 2)), axis=0)
    lcl = SimpleNamespace(target_lane=lane, target_velocity=4.0)
    trees = []
    for M in (25, 40, 7):
        par = np.concatenate([[-1], rng.integers(0, np.arange(1, M))]).astype(np.int32)
        xs = rng.normal(size=(M, 6)) * np.array([20, 20, 2, 0.3, 1, 0.1]) + np.array([30, 30, 4, 0, 0, 0])
        trees.append(to_traj_tree(dict(parent=par), xs[0] * 0.9, xs, rng.normal(size=(M, 2))))
    got = pl.evaluate_traj_trees(lcl, trees)
    want = [pl.evaluate_traj_tree(lcl, t) for t in trees]
    assert np.allclose(got, want, rtol=1e-13, atol=0) and int(np.argmin(got)) == int(np.argmin(want))
    for t in trees:
        t._arrays = None
    assert pl.evaluate_traj_trees(lcl, trees) == want


from oracle import ref_harness as _rh  # noqa: E402


@pytest.mark.reference
@pytest.mark.skipif(not _rh.available(), reason="reference tree not present")
def test_planner_helpers_match_reference_live():
    """Build container only: MINDPlanner.resample_target_lane and evaluate_traj_tree (planner.py:147-198) of the imported
    reference on random inputs against this package's vectorised versions."""
    from types import SimpleNamespace
    from mind_amd.planners.mind.trajectory_tree import to_traj_tree
    m = _rh.ref_modules()
    RefPlanner = m["planners.mind.planner"].MINDPlanner
    RefTree, RefNode = m["planners.basic.tree"].Tree, m["planners.basic.tree"].Node
    ref, mine = RefPlanner.__new__(RefPlanner), MINDPlanner.__new__(MINDPlanner)
    rng = np.random.default_rng(11)
    for trial in range(6):
        P = int(rng.integers(3, 40))
        dtype = np.float32 if trial % 2 else np.float64
        lane = np.cumsum(rng.uniform(0.3, 6.0, (P, 2)), axis=0).astype(dtype)
        info = [rng.integers(0, 2, P).astype(np.float32), np.eye(3, dtype=np.float32)[rng.integers(0, 3, P)],
                np.eye(3, dtype=np.float32)[rng.integers(0, 3, P)], np.eye(3, dtype=np.float32)[rng.integers(0, 3, P)],
                rng.integers(0, 2, P).astype(np.float32), rng.integers(0, 2, P).astype(np.float32)]
        lcl = SimpleNamespace(target_lane=lane, target_lane_info=info, target_velocity=float(rng.uniform(2, 9)))
        wl, wi = ref.resample_target_lane(lcl)
        gl, gi = mine.resample_target_lane(lcl)
        assert np.asarray(gl).dtype == np.asarray(wl).dtype and np.array_equal(gl, wl)
        assert len(gi) == len(wi) and all(np.array_equal(a, b) for a, b in zip(gi, wi))
        # evaluation of a random trajectory tree
        M = int(rng.integers(5, 60))
        par = np.concatenate([[-1], rng.integers(0, np.arange(1, M))]).astype(np.int32)
        xs = rng.normal(size=(M, 6)) * np.array([20, 20, 2, 0.3, 1, 0.1]) + np.array([30, 30, 4, 0, 0, 0])
        us = rng.normal(size=(M, 2))
        x0 = xs[0] * 0.9
        tt = to_traj_tree(dict(parent=par), x0, xs, us)
        rt = RefTree()
        rt.add_node(RefNode(-1, None, [x0, np.zeros(2)]))
        for k in range(M):
            rt.add_node(RefNode(k, int(par[k]), [xs[k], us[k]]))
        want = ref.evaluate_traj_tree(lcl, rt)
        assert abs(mine.evaluate_traj_tree(lcl, tt) - want) <= 1e-12 * abs(want)
        assert abs(mine.evaluate_traj_trees(lcl, [tt])[0] - want) <= 1e-12 * abs(want)


@pytest.mark.reference
@pytest.mark.skipif(not _rh.available(), reason="reference tree not present")
def test_observation_bookkeeping_matches_reference_live():
    """Build container only: MINDPlanner.update_observation (planner.py:65-95: 50-frame sliding tracks, dummy states for
    agents missing from a frame) followed by get_agent_trajectories (utils.py:245-342) of the imported reference, fed with a
    random appear / disappear pattern over 75 frames -- tracks, flags, padding, type one-hots and agent order after every
    frame from the tenth on."""
    import torch as _torch
    from types import SimpleNamespace as NS
    m = _rh.ref_modules()
    ref = m["planners.mind.planner"].MINDPlanner.__new__(m["planners.mind.planner"].MINDPlanner)
    ref.agent_obs, ref.obs_len = {}, 50
    mine = MINDPlanner.__new__(MINDPlanner)
    mine.agent_obs, mine.obs_len = {}, 50
    rng = np.random.default_rng(21)
    ids = ["AV"] + [str(100 + i) for i in range(9)]
    types = [list(_rh.ObjectType)[i % 7] for i in range(10)]
    present = {i: True for i in ids}
    for frame in range(75):
        for i in ids[1:]:
            if rng.random() < 0.12:
                present[i] = not present[i]
        def obs(i):
            s = np.array([rng.normal() * 40, rng.normal() * 40, abs(rng.normal()) * 5, rng.uniform(-3, 3)])
            return NS(id=i, type=types[ids.index(i)], state=s, timestep=frame)
        ego = obs("AV")
        exo = [obs(i) for i in ids[1:] if present[i]]
        rng.shuffle(exo)                                                       # first-appearance order defines the agent order
        lcl = NS(ego_agent=ego, exo_agents=exo)
        ref.update_observation(lcl)
        mine.update_observation(lcl)
        assert list(ref.agent_obs) == list(mine.agent_obs)
        if frame < 10:
            continue
        want = m["planners.mind.utils"].get_agent_trajectories(ref.agent_obs, _torch.device("cpu"))
        got = U.get_agent_trajectories(mine.agent_obs)
        assert got[5] == want[5] and got[6] == want[6], frame
        for g_, w_ in zip(got[:5], want[:5]):
            w_ = w_.numpy()
            assert g_.dtype == w_.dtype and g_.shape == w_.shape and np.array_equal(g_, w_), frame


def test_child_scene_views_equal_the_reference_concatenation():
    """ChildScene (a kept mode that references its parent's history and a view of its own rows) against the arrays
    prune_merge builds in the reference (scenario_tree.py:396-412: parent history ++ 60 predicted steps, truncated to
    seq_len): every accessor the AIME bookkeeping uses -- rows(), window6(), the by-name materialisation, trim() -- for
    parents of 50 steps (root / re-based nodes) and of another length."""
    from mind_amd.planners.mind.scenario_tree import ChildScene, hist_len, hist_rows, hist_trim
    rng = np.random.default_rng(5)
    for n_hist, L in ((50, 100), (37, 100), (50, 80)):
        a = 7
        parent = {"TRAJS_POS_HIST": rng.standard_normal((a, n_hist, 2)).astype(np.float32),
                  "TRAJS_VEL_HIST": rng.standard_normal((a, n_hist, 2)).astype(np.float32),
                  "TRAJS_ANG_HIST": rng.standard_normal((a, n_hist)).astype(np.float32),
                  "TRAJS_COV_HIST": rng.random((a, n_hist, 1)).astype(np.float32)}
        new = rng.standard_normal((a, 60, 6)).astype(np.float32)
        want = {"TRAJS_POS_HIST": np.concatenate([parent["TRAJS_POS_HIST"], new[:, :, 0:2]], 1)[:, :L],
                "TRAJS_VEL_HIST": np.concatenate([parent["TRAJS_VEL_HIST"], new[:, :, 2:4]], 1)[:, :L],
                "TRAJS_ANG_HIST": np.concatenate([parent["TRAJS_ANG_HIST"], new[:, :, 4]], 1)[:, :L],
                "TRAJS_COV_HIST": np.concatenate([parent["TRAJS_COV_HIST"], new[:, :, 5:6]], 1)[:, :L]}
        c = ChildScene({"CUR_T": 0, "END_T": 50}, parent, new, L)
        assert hist_len(c) == min(n_hist + 60, L) and "TRAJS_POS_HIST" in c and "NOPE" not in c
        for k, w in want.items():
            for lo, hi in ((0, 10), (n_hist - 5, n_hist + 7), (n_hist, n_hist + 20), (60, 100), (0, 200)):
                assert np.array_equal(hist_rows(c, k, lo, hi), w[:, lo:hi]), (k, lo, hi)
        w6 = c.window6(30, 80)
        assert np.array_equal(w6[:, :, 0:2], want["TRAJS_POS_HIST"][:, 30:80]) and np.array_equal(w6[:, :, 4], want["TRAJS_ANG_HIST"][:, 30:80])
        assert np.array_equal(w6[:, :, 5:6], want["TRAJS_COV_HIST"][:, 30:80]) and np.array_equal(w6[:, :, 2:4], want["TRAJS_VEL_HIST"][:, 30:80])
        for k, w in want.items():                       # by-name access builds exactly the reference's array
            assert np.array_equal(c[k], w) and c[k].dtype == np.float32
        hist_trim(c, 72)                                # update_obser's truncation, also of already materialised arrays
        assert hist_len(c) == min(72, min(n_hist + 60, L))
        for k, w in want.items():
            assert np.array_equal(c[k], w[:, :72]) and np.array_equal(hist_rows(c, k, 22, 72), w[:, 22:72])
        with pytest.raises(KeyError):
            c["MISSING"]


def test_device_window_bookkeeping_and_lazy_host_windows():
    """_update_obser_device_windows (re-basing windows cut out on the device): with a stand-in runtime that only records what it is
    asked, the request must name the right parent slots / row offsets / durations, and the DevScenes it returns must hold, lazily,
    exactly the windows the host path uploads (ChildScene.window6), the right cov_last and history length -- and chain: a
    grandchild built on a lazy DevScene sees the same parent history as one built on uploaded windows."""
    import torch
    from types import SimpleNamespace
    from mind_amd.planners.mind.configs.planning._base import ScenTreeCfg
    from mind_amd.planners.mind.scenario_tree import ChildScene, DevScene, ScenarioTreeGenerator, cov_last_of, hist_len
    rng = np.random.default_rng(9)
    a, o = 5, 50
    gen = ScenarioTreeGenerator(torch.device("cpu"), None, 50, 60, ScenTreeCfg())
    gen.target_lane = np.stack([np.arange(40, dtype=np.float32), np.zeros(40, np.float32)], 1)
    gen.target_lane_info = np.zeros((40, 12), np.float32)
    gen.lane_graph = {"lane_ctrs": np.zeros((3, 2), np.float32), "lane_vecs": np.ones((3, 2), np.float32)}
    types = np.zeros((a, 50, 7), np.float32)
    calls = []

    class FakeRT:
        _rebase_gen = 7

        def aime_rebase(self, pos, ang, vel, types_, lc, lv, tl, ti, time_ahead=5.0, dev_src=None, **kw):
            calls.append(dev_src)
            S = len(dev_src["parent_slot"])
            self._rebase_gen += 1
            return {"frames": torch.zeros(S, 28), "gen": self._rebase_gen}
    rt = FakeRT()
    prev_dev = {"gen": 7, "a": a, "l": 3}
    parents = []
    for g in range(3):                                   # three parents re-based by "the previous call" (host windows known)
        w6 = rng.standard_normal((a, 50, 6)).astype(np.float32)
        p = DevScene(gen, prev_dev, g, {"TRAJS_POS_HIST": w6[:, :, 0:2], "TRAJS_VEL_HIST": w6[:, :, 2:4], "TRAJS_ANG_HIST": w6[:, :, 4],
                                        "TRAJS_COV_HIST": w6[:, :, 5:6], "TRAJS_TYPE": types, "SCEN_ID": f"p{g}"})
        p._win6 = w6
        parents.append(p)
    rows_dev = object()                                  # stands for the device tensor of the round's kept rows
    curs, want = [], []
    for j, (pg, dur) in enumerate(((2, 4), (0, 30), (2, 60), (1, 1))):
        new = rng.standard_normal((a, 60, 6)).astype(np.float32)
        c = ChildScene({"CUR_T": 0, "END_T": dur, "SCEN_PROB": np.float32(0.5), "SCEN_ID": f"c{j}", "PARENT_ID": f"p{pg}", "TRAJS_TYPE": types,
                        "TRAJS_TID": None, "TRAJS_CAT": None}, parents[pg], new, 110)
        c.dev_rows, c.row0 = rows_dev, j * a
        curs.append(c)
        want.append(np.concatenate([parents[pg]._win6, new], axis=1)[:, dur:dur + 50])
    out = gen._update_obser_device_windows(curs, rt)
    assert out is not None and len(calls) == 1
    src = calls[0]
    assert src["rows"] is rows_dev and src["gen"] == 7 and src["a"] == a
    assert list(src["parent_slot"]) == [2, 0, 2, 1] and list(src["row0"]) == [0, a, 2 * a, 3 * a] and list(src["dur"]) == [4, 30, 60, 1]
    for (obs, cur), w, c in zip(out, want, curs):
        assert cur is c and obs.lazy is not None and hist_len(obs) == 50 and not dict.__contains__(obs, "TRAJS_POS_HIST")
        assert np.array_equal(cov_last_of(obs), w[:, -1, 5])
        assert "TRAJS_POS_HIST" in obs and np.array_equal(obs["TRAJS_POS_HIST"], w[:, :, 0:2]) and np.array_equal(obs["TRAJS_ANG_HIST"], w[:, :, 4])
        assert np.array_equal(obs["TRAJS_VEL_HIST"], w[:, :, 2:4]) and np.array_equal(obs["TRAJS_COV_HIST"], w[:, :, 5:6])
        assert obs["CUR_T"] == c["END_T"] and obs.g == curs.index(c)
    # a grandchild of a lazily held scene: its window is cut out of the same parent history
    lazy_parent = out[0][0]
    assert gen._update_obser_device_windows(curs[:1], rt) is None          # the parents' arena is two calls old by now
    new = rng.standard_normal((a, 60, 6)).astype(np.float32)
    gc = ChildScene({"CUR_T": 4, "END_T": 14}, lazy_parent, new, 110)
    assert hist_len(gc) == 110 and np.array_equal(gc.window6(10, 60), np.concatenate([want[0], new], axis=1)[:, 10:60])
    # not applicable: a parent that is not a re-based scene, or a stale generation -> None (the caller uploads the windows)
    rt._rebase_gen = 99
    assert gen._update_obser_device_windows(curs, rt) is None


def test_the_exchange_on_a_shared_context_follows_the_planner_that_plans():
    """A context (per-thread runtime) outlives the planner that attached a process group to it (mind_set_exchange).  Before its native
    plan a scenario-tree generator makes the context's exchange its own: a planner without a shard takes a previous planner's group
    off, a sharded planner puts its group back (bench.py's multi-rank line plans `tree_sharded` and then `tree_replicas` on one
    context: the replicas must not shard)."""
    calls = []

    class Lib:
        def mind_set_exchange(self, ctx, rank, world, fn, user, force):
            calls.append((rank, world, bool(fn), force))
            return 0

        def mind_last_error_string(self, ctx):
            return b""

    class Rt:
        lib, ctx = Lib(), None

    class Sh:
        native = True

        def attach(self, rt):
            calls.append("attach")
            rt._exchange_owner = self

    rt = Rt()
    g = ScenarioTreeGenerator.__new__(ScenarioTreeGenerator)
    g.shard = None
    g._sync_exchange(rt)                     # nothing attached, nothing wanted: no call
    assert calls == []
    g.shard = sh = Sh()
    g._sync_exchange(rt)
    assert calls == ["attach"] and rt._exchange_owner is sh
    g._sync_exchange(rt)                     # already this planner's
    assert calls == ["attach"]
    g2 = ScenarioTreeGenerator.__new__(ScenarioTreeGenerator)
    g2.shard = None
    g2._sync_exchange(rt)                    # the next planner on the context plans alone: rank 0 of 1, no callback
    assert calls[-1] == (0, 1, False, 0) and rt._exchange_owner is None
    g._sync_exchange(rt)
    assert calls[-1] == "attach"


def test_lazy_tree_is_a_tree_once_somebody_looks():
    """LazyTree (planners/basic/tree.py): the native planner's trees are built on first access; every Tree method then sees the same
    container the eager construction gives, the builder runs once, and the root key is available without building."""
    from mind_amd.planners.basic.tree import LazyTree
    built = []

    def fill(t):
        built.append(1)
        t.add_node(Node("r", None, 0))
        t.add_node(Node("a", "r", 1))
        t.add_node(Node("b", "r", 2))
        t.add_node(Node("c", "a", 3))

    eager = Tree()
    fill(eager)
    built.clear()
    lz = LazyTree(fill, root="r")
    assert lz.get_root_key() == "r" and not built                    # known without building
    assert lz.size() == 4 and built == [1]
    assert list(lz.nodes) == list(eager.nodes) and lz.leaves == eager.leaves == ["b", "c"]
    assert [n.key for n in lz.retrieve_nodes_to_root("c")] == ["c", "a", "r"] and lz.get_root().key == "r"
    assert lz.get_node("a").children_keys == ["c"] and lz.get_node("c").depth == 2
    lz.add_node(Node("d", "b", 4))
    assert lz.leaves == ["c", "d"] and built == [1]
    with pytest.raises(KeyError):
        lz.get_node("zz")
    lz2 = LazyTree(fill)                                               # root unknown until built
    assert lz2.get_root_key() == "r" and lz2.get_leaf_keys() == ["b", "c"]
