"""The closed loop behind one native call per step (mind_loop_*, mind_amd/native_loop.py) against the Python steps of ClosedLoopSim
(mind_amd/closed_loop.py: the reference's simulator.py:51-107 / agent.py:255-331 / planner.py:50-145 in this repo's Python form): the same
kernels on the same windows, so EVERYTHING must be bit-identical cycle by cycle -- ego state and control, chosen tree, candidate costs,
the returned scenario / trajectory trees, the running counters -- across an episode restart, and after the loop is handed back to
the Python steps in the middle of a run."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _make(scene, native, episode_plans=None, speculative=False):
    sys.path.insert(0, ROOT)
    from bench import BRANCHING_WEIGHTS, WORKLOADS
    from mind_amd.closed_loop import ClosedLoopSim
    from mind_amd.planners.mind.planner import MINDPlanner
    from mind_amd.scene_io import ReplayWorld, scene_fixture_path
    import json
    cfg = os.path.join(ROOT, "mind_amd", "planners", "mind", "configs", "synthetic.json")
    wkw = dict(WORKLOADS[scene])
    w = ReplayWorld.from_scene_file(scene_fixture_path(wkw["scene"]))
    cfg = dict(json.load(open(cfg)), planning_config="planners.mind.configs.planning." + wkw["scene"], ckpt_path=BRANCHING_WEIGHTS)
    pl = MINDPlanner(cfg)
    pl.traj_tree_opt.speculative = speculative
    sim = ClosedLoopSim(w, pl, episode_plans=episode_plans, native=native)
    sim.run_until(sim.enable_time)
    return pl, sim


def _flat(trees):
    out = []
    for t in trees:
        for k, n in t.nodes.items():
            out.append((k, n.parent_key) + tuple(np.asarray(n.data[i]).copy() for i in range(4)))
    return out


def _snapshot(pl, sim):
    scen, traj = sim.last_result
    tt = traj[0]
    return dict(state=np.array(sim.state), ctrl=np.array(sim.ctrl), t=sim.sim_time, n_steps=sim.n_steps, n_plans=sim.n_plans,
                best=pl.timing["best_traj_idx"], costs=np.array(pl.timing["tree_costs"]), n_exp=pl.timing["nodes_expanded"],
                scen=_flat(scen), traj=(np.array(tt._arrays[0]), np.array(tt._arrays[1])),
                traj_nodes=[(k, n.parent_key, np.array(n.data[0]), np.array(n.data[1])) for k, n in tt.nodes.items()],
                expanded=pl.scen_tree_gen.n_expanded, counters=dict(pl.traj_tree_opt.counters), plans=pl.timing_sum["plans"])


def _same(a, b, cycle):
    for k in ("state", "ctrl", "costs"):
        assert np.array_equal(a[k], b[k]), (cycle, k, a[k], b[k])
    for k in ("t", "n_steps", "n_plans", "best", "n_exp", "expanded", "plans"):
        assert a[k] == b[k], (cycle, k, a[k], b[k])
    for k in ("solves", "iterations", "node_iterations", "node_iterations_exo"):
        assert a["counters"].get(k, 0) == b["counters"].get(k, 0), (cycle, k)
    assert len(a["scen"]) == len(b["scen"])
    for x, y in zip(a["scen"], b["scen"]):
        assert x[0] == y[0] and x[1] == y[1], (cycle, x[0], y[0])
        for u, v in zip(x[2:], y[2:]):
            assert u.dtype == v.dtype and u.shape == v.shape and np.array_equal(u, v), (cycle, x[0])
    assert np.array_equal(a["traj"][0], b["traj"][0]) and np.array_equal(a["traj"][1], b["traj"][1]), cycle
    assert len(a["traj_nodes"]) == len(b["traj_nodes"])
    for x, y in zip(a["traj_nodes"], b["traj_nodes"]):
        assert x[0] == y[0] and x[1] == y[1] and np.array_equal(x[2], y[2]) and np.array_equal(x[3], y[3]), (cycle, x[0])


@pytest.mark.parametrize("scene", ["demo_1", "demo_2", "demo_3", "demo_4"])
def test_native_loop_equals_the_python_steps(scene):
    """14 planning cycles with episodes of 6 (two restarts: windows cleared, history replayed to the enable time), one cycle per call"""
    pa, sa = _make(scene, None, episode_plans=6)
    pb, sb = _make(scene, False, episode_plans=6)
    assert sa._native is not None and sa.native_reason is None and sb._native is None
    assert np.array_equal(sa.state, sb.state) and sa.sim_time == sb.sim_time and sa.n_steps == sb.n_steps
    multi = 0
    for cycle in range(14):
        na = sa.run_plans(1)
        a = _snapshot(pa, sa)              # (the two planners share the thread's context: a plan's tables are read before the other one plans)
        nb = sb.run_plans(1)
        b = _snapshot(pb, sb)
        assert na == nb, (cycle, na, nb)
        _same(a, b, cycle)
        multi += len(a["costs"]) > 1
    assert sa._native is not None and sa.n_episodes == sb.n_episodes == 2
    assert multi >= 4                      # the branching weights really offer several candidate trees
    assert len(pa.agent_obs) == 0          # the windows live in the library ...
    # ... and come back as the Python driver keeps them
    sa._native.hand_back()
    assert sa._native is None and list(pa.agent_obs) == list(pb.agent_obs)
    for k in pa.agent_obs:
        ta, tb = pa.agent_obs[k], pb.agent_obs[k]
        assert len(ta.object_states) == len(tb.object_states) and np.array_equal(ta._arr, tb._arr), k
        assert [tuple(s)[:2] for s in ta.object_states] == [tuple(s)[:2] for s in tb.object_states], k


def test_native_loop_step_by_step_and_in_one_call():
    """ClosedLoopSim.step (one simulator step per call) and run_plans(5) (one call for five cycles) walk the same loop"""
    pa, sa = _make("demo_1", None)
    pb, sb = _make("demo_1", None)
    pc, sc = _make("demo_1", False)
    planned = 0
    while planned < 5:
        planned += bool(sa.step())
    a = _snapshot(pa, sa)
    sb.run_plans(5)
    b = _snapshot(pb, sb)
    sc.run_plans(5)
    c = _snapshot(pc, sc)
    assert sa._native is not None and sb._native is not None
    _same(a, c, "step")
    _same(b, c, "run_plans(5)")
    assert pb.timing_sum["plans"] == 5 and pb.scen_tree_gen.n_native_plans == 5
    # the plan tables are the context's: once another planner has planned on it, the loop's last plan can no longer be materialised
    from mind_amd._lib import MindError
    sa._native._result = None
    with pytest.raises(MindError, match="planned again"):
        sa.last_result


def test_loop_is_handed_back_when_the_planner_changes_under_it():
    """five native cycles, then the host featuriser is selected (device_root = False): the loop's windows move into planner.agent_obs and the
    Python steps continue from the same state -- equal to a run that took the Python steps all along with the same switch"""
    pa, sa = _make("demo_2", None)
    pb, sb = _make("demo_2", False)
    for cycle in range(10):
        if cycle == 5:
            pa.scen_tree_gen.device_root = False
            pb.scen_tree_gen.device_root = False
        sa.run_plans(1)
        a = _snapshot(pa, sa)
        sb.run_plans(1)
        assert (sa._native is not None) == (cycle < 5)
        _same(a, _snapshot(pb, sb), cycle)
    assert list(pa.agent_obs) == list(pb.agent_obs)


def test_drivers_that_step_in_halves_take_the_loop_over():
    """step_begin / step_end (mind_amd/pipelined.py, fused.py drive the simulator in halves) hand the loop back at once"""
    pa, sa = _make("demo_3", None)
    pb, sb = _make("demo_3", False)
    sa.run_plans(2); sb.run_plans(2)
    snaps = []
    for pl, sim in ((pa, sa), (pb, sb)):
        for _ in range(12):
            lcl = sim.step_begin()
            sim.step_end(pl.plan(lcl) if lcl is not None else None)
        snaps.append(_snapshot(pl, sim))
    assert sa._native is None
    _same(snaps[0], snaps[1], "halves")


def test_native_loop_refuses_what_it_does_not_cover():
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    pl, sim, w = make_closed_loop(dict(WORKLOADS["demo1"]), native=None)           # scripted modes on a synthetic world
    assert sim._native is None and "network" in sim.native_reason
    from mind_amd.closed_loop import ClosedLoopSim
    with pytest.raises(RuntimeError, match="native=True"):
        ClosedLoopSim(w, pl, episode_plans=24, native=True)


@pytest.mark.parametrize("scene", ["demo_1", "demo_2", "demo_3", "demo_4"])
def test_native_loop_against_the_references_own_closed_loop(scene):
    """The reference's simulator loop on its four recorded scenes (tests/golden/demo_plans.npz, captured from the imported reference: trigger
    steps, AIME branch ids, agent and ego trajectories of the first four planning cycles, final state and control) -- the check of
    tests/test_gpu_plan.py::test_recorded_demo_scenes_match_reference_closed_loop, driven through mind_loop_advance one simulator step at a time."""
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS, make_closed_loop
    from test_gpu_plan import _closed_loop_against_demo_plans
    pl, sim, w = make_closed_loop(dict(WORKLOADS[scene]), scripted=False, native=None)
    assert sim._native is not None, sim.native_reason
    _closed_loop_against_demo_plans(scene, pl, sim, w)


@pytest.mark.parametrize("scene", ["demo_1", "demo_4"])
def test_native_loop_whole_episode_equals_the_python_steps(scene):
    """all 60 planning cycles of the reference's episode (t = 4.0 .. 9.9 s): ego state, control, chosen tree and candidate costs after every
    cycle, the native loop ten cycles per call"""
    pa, sa = _make(scene, None, episode_plans=60)
    pb, sb = _make(scene, False, episode_plans=60)
    for block in range(6):
        sa.run_plans(10)
        a = (np.array(sa.state), np.array(sa.ctrl), pa.timing["best_traj_idx"], np.array(pa.timing["tree_costs"]), sa.n_steps, pa.scen_tree_gen.n_expanded)
        sb.run_plans(10)
        b = (np.array(sb.state), np.array(sb.ctrl), pb.timing["best_traj_idx"], np.array(pb.timing["tree_costs"]), sb.n_steps, pb.scen_tree_gen.n_expanded)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3]) and a[4:] == b[4:], block
    assert sa._native is not None and sa.n_plans == 60


@pytest.mark.parametrize("scene", ["demo_2", "demo_4"])
def test_native_loop_with_the_speculative_warm_start(scene, monkeypatch):
    """MIND_NATIVE_SPECULATE=1: the loop runs the optimizer's speculative warm start itself (warm-start fits of the previous plan's tree shapes
    on a second context beside the AIME rounds; hits run the full fit only, misses both fits beside them, the back-off schedule of
    TrajectoryTreeOptimizer).  Same kernels on the same inputs: every cycle is the same bits as the Python steps WITHOUT speculation, and
    the counters show that fits were speculated and used."""
    monkeypatch.setenv("MIND_NATIVE_SPECULATE", "1")
    pa, sa = _make(scene, None, episode_plans=30, speculative=True)
    monkeypatch.delenv("MIND_NATIVE_SPECULATE")
    pb, sb = _make(scene, False, episode_plans=30, speculative=False)
    assert sa._native is not None and sa._native.speculative
    for cycle in range(30):
        sa.run_plans(1)
        a = _snapshot(pa, sa)
        sb.run_plans(1)
        b = _snapshot(pb, sb)
        a["counters"] = b["counters"] = {}            # (the speculation's own counters differ by design)
        _same(a, b, cycle)
    cn = pa.traj_tree_opt.counters
    print(f"{scene}: {cn['warm_speculated']} warm-start fits speculated, {cn['warm_hits']} used, over 30 cycles")
    assert cn["warm_speculated"] > 0


def test_copy_and_table_knobs_leave_every_cycle_bit_identical():
    """The copies taken off the cycle's critical path are pure data movement: uploads as kernels that read the page-locked staging (pl_upload),
    results written to the host staging by k_ilqr / k_aime_branch themselves, small index tables read from the staging or passed in the
    kernel arguments (AimeSmall), candidates priced as soon as their tree is marked complete, the tree-iLQR kernel's derivative
    speculator, the pruning decisions and branch-time bits from one launch (k_aime_select_branch).  With all of them off (copies, uploaded tables, the batch evaluation behind the finish, the master's own derivative pass)
    the native loop must walk exactly the same cycles -- on a scene whose plans branch."""
    knobs = {"upload_kernel_max": (0, 1 << 20), "ilqr_host_out_max": (0, 4096), "dec_mirror": (0, 1), "tab_host_max": (0, 4096), "tab_small": (0, 1),
             "early_eval": (0, 1), "ilqr_spec_deriv": (0, 1), "glue_fused": (0, 1)}
    runs = []
    rt = None
    try:
        for arm in (0, 1):
            pl, sim = _make("demo_4", None, episode_plans=5)
            rt = pl.network.rt
            for k, v in knobs.items():
                rt.set_tuning(k, v[arm])
            assert sim._native is not None
            snaps = []
            for cycle in range(8):
                sim.run_plans(1)
                snaps.append(_snapshot(pl, sim))
            runs.append(snaps)
            if arm == 1:
                asked, hits = (int(v) for v in rt.debug_read("il_spec"))
                assert asked > 0 and 0 < hits <= asked          # (last launch of the run: the speculator's set became the nominal one)
    finally:
        if rt is not None:
            for k, v in knobs.items():
                rt.set_tuning(k, v[1])
    assert sum(len(s["costs"]) > 1 for s in runs[0]) >= 2
    for cycle, (a, b) in enumerate(zip(*runs)):
        _same(a, b, cycle)
