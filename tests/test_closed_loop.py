"""f1: closed-loop driver.  CPU: plant and trigger logic (plant pinned against the imported reference
in the build container); GPU: a short closed loop with the real planner."""
import numpy as np
import pytest

from mind_amd.closed_loop import ClosedLoopSim, kine_propagate
from mind_amd.synth import SynthWorld
from oracle import ref_harness as rh


def test_kine_propagate_values():
    s = kine_propagate(np.array([1.0, 2.0, 3.0, 0.5]), np.array([1.0, 0.1]), 0.02, 3.0, 15.0, np.deg2rad(45.0))
    exp = np.array([1.0 + 3.0 * np.cos(0.5) * 0.02, 2.0 + 3.0 * np.sin(0.5) * 0.02, 3.02, 0.5 + 3.0 / 3.0 * np.tan(0.1) * 0.02])
    assert np.allclose(s, exp, atol=1e-15)
    s = kine_propagate(np.array([0, 0, 14.99, 0.0]), np.array([9.0, 2.0]), 0.02, 3.0, 15.0, np.deg2rad(45.0))
    assert s[2] == 15.0 and np.isclose(s[3], 14.99 / 3.0 * np.tan(np.deg2rad(45.0)) * 0.02)     # clipped a, delta, v


@pytest.mark.reference
@pytest.mark.skipif(not rh.available(), reason="reference tree not present")
def test_kine_propagate_matches_reference():
    rh.install()
    from common.kinematics import kine_propagate as ref
    rng = np.random.default_rng(0)
    for _ in range(50):
        st, ct = rng.normal(size=4) * [10, 10, 5, 1], rng.normal(size=2) * [4, 0.5]
        assert np.array_equal(ref(st, ct, 0.02, 3.0, 15.0, np.deg2rad(45.0)), kine_propagate(st, ct, 0.02, 3.0, 15.0, np.deg2rad(45.0)))


class _StubPlanner:
    def __init__(self):
        self.obs, self.plans, self.lane = 0, [], None

    def update_target_lane(self, lane):
        self.lane = lane

    def update_observation(self, lcl):
        self.obs += 1

    def update_state_ctrl(self, s, c):
        self.sc = (np.array(s), np.array(c))

    def plan(self, lcl):
        self.plans.append(round(lcl.ego_agent.timestep * 0.1, 3))
        return True, np.array([0.5, 0.0]), None


def test_trigger_schedule_matches_reference_counts():
    """500 steps of 0.02 s: 100 observation updates (10 Hz), 60 plans for t in [4, 10)."""
    w = SynthWorld(n_agents=3, n_lanes=2, n_segs=6, seed=1)
    p = _StubPlanner()
    sim = ClosedLoopSim(w, p)
    for _ in range(500):
        sim.step()
    assert p.obs == 100 and len(p.plans) == 60 and sim.n_plans == 60
    assert abs(sim.state[2] - (w.agent_speed[0] + 0.5 * 0.02 * 300)) < 0.2 or sim.state[2] == 15.0    # accelerating ego


class _HookPlanner(_StubPlanner):
    """a planner that calls its idle hook once per plan, as MINDPlanner does while the device computes; records every observation"""

    def __init__(self, call_hook):
        super().__init__()
        self.call_hook, self.seen = call_hook, []

    def to_object_state(self, agent):
        return ("state", agent.id, tuple(np.asarray(agent.state).tolist()), agent.timestep)

    def update_observation(self, lcl):
        super().update_observation(lcl)
        self.seen.append([(a.id, tuple(np.asarray(a.state).tolist()), a.timestep, getattr(a, "obj_state", None)) for a in lcl.exo_agents])

    def plan(self, lcl):
        res = super().plan(lcl)
        hook = getattr(self, "idle_hook", None)
        if self.call_hook and hook is not None:
            hook()
        return res


def test_observation_built_ahead_in_the_idle_hook_is_the_observation_that_comes():
    """ClosedLoopSim registers MINDPlanner.idle_hook: the replayed agents of the NEXT planning step (and their ObjectStates) are built
    while the device computes and used only if the simulator reaches exactly the time they were built for -- every observation the
    planner sees must be the one it would have seen without the hook, episode restarts included."""
    runs = []
    for call_hook in (True, False):
        w = SynthWorld(n_agents=5, n_lanes=2, n_segs=6, seed=2)
        p = _HookPlanner(call_hook)
        sim = ClosedLoopSim(w, p, episode_plans=20)
        assert callable(getattr(p, "idle_hook", None))
        sim.run_plans(45)                                   # two episode restarts on the way
        runs.append(p)
    a, b = runs
    assert len(a.seen) == len(b.seen) > 45 and a.plans == b.plans
    used = 0
    for fa, fb in zip(a.seen, b.seen):
        assert [x[:3] for x in fa] == [x[:3] for x in fb]
        for x in fa:
            if x[3] is not None:                             # built ahead: the attached ObjectState is the planner's own conversion
                assert x[3] == ("state", x[0], x[1], x[2])
                used += 1
    assert used > 100 and all(x[3] is None for f in b.seen for x in f)


class _TwoHalfPlanner(_StubPlanner):
    """plan() in two halves, logging the order in which a driver calls them"""

    def __init__(self, name, log):
        super().__init__()
        self.name, self.log = name, log

    def plan_begin(self, lcl):
        self.log.append(("begin", self.name))
        return lcl

    def plan_end(self, begun):
        self.log.append(("end", self.name))
        return _StubPlanner.plan(self, begun)

    def plan(self, lcl):
        return self.plan_end(self.plan_begin(lcl))


def test_pipelined_driver_keeps_one_plan_in_flight_and_plans_what_the_scenes_plan_alone():
    """mind_amd.pipelined.PipelinedClosedLoops (CPU, stub planners): scene i + 1's plan_begin is called before scene i's plan_end -- one
    plan in flight --, every scene gets exactly n plans per run_plans(n), at the simulator times it plans at alone (episode restarts
    included), and a single scene degenerates to begin / end pairs."""
    from mind_amd.pipelined import PipelinedClosedLoops
    log = []
    sims = [ClosedLoopSim(SynthWorld(n_agents=3, n_lanes=2, n_segs=6, seed=s), _TwoHalfPlanner("s%d" % s, log), episode_plans=7) for s in (1, 2, 3)]
    pc = PipelinedClosedLoops(sims)
    steps = pc.run_plans(5) + pc.run_plans(6)               # 11 plans per scene: one episode restart each
    assert [s.n_plans for s in sims] == [11, 11, 11] and steps == sum(s.n_steps for s in sims)
    alone = ClosedLoopSim(SynthWorld(n_agents=3, n_lanes=2, n_segs=6, seed=1), _StubPlanner(), episode_plans=7)
    alone.run_plans(11)
    assert sims[0].planner.plans == alone.planner.plans and sims[0].n_steps == alone.n_steps and np.array_equal(sims[0].state, alone.state)
    # order: never two ends in a row without a begin between them except when a run drains, every begin of scene X is followed by X's end
    # only after the NEXT scene's begin (within a run of run_plans)
    first = log[:2 * 3 * 5]
    assert first[0] == ("begin", "s1") and first[1] == ("begin", "s2") and first[2] == ("end", "s1") and first[3] == ("begin", "s3") and first[4] == ("end", "s2")
    open_ = []
    for kind, name in log:
        if kind == "begin":
            open_.append(name)
            assert len(open_) <= 2                          # one plan in flight behind the one being started
        else:
            assert open_ and open_[0] == name               # plans complete in the order they were started
            open_.pop(0)
    assert not open_
    log.clear()
    one = PipelinedClosedLoops([ClosedLoopSim(SynthWorld(n_agents=3, n_lanes=2, n_segs=6, seed=4), _TwoHalfPlanner("x", log))])
    one.run_plans(3)
    assert log == [("begin", "x"), ("end", "x")] * 3
    with pytest.raises(TypeError):
        PipelinedClosedLoops([ClosedLoopSim(SynthWorld(n_agents=3, n_lanes=2, n_segs=6, seed=5), _StubPlanner())])


class _ThreePiecePlanner(_TwoHalfPlanner):
    """plan() in three pieces whose device work takes a number of readiness polls to finish (a stand-in for the native AIME plan on a
    library thread and the contingency solves on the scene's stream)"""

    def __init__(self, name, log, aime_polls, solve_polls):
        super().__init__(name, log)
        self.aime_polls, self.solve_polls, self.left = aime_polls, solve_polls, 0

    def plan_start(self, lcl):
        self.log.append(("start", self.name))
        self.left = self.aime_polls
        return lcl

    def plan_started_ready(self, started):
        self.left -= 1
        return self.left < 0

    def plan_begin_finish(self, started):
        self.log.append(("mid", self.name, self.left < 0))           # (collected without waiting?)
        self.left = self.solve_polls
        return started

    def plan_end_ready(self, begun):
        self.left -= 1
        return self.left < 0

    def plan_end_piece(self, begun):
        self.log.append(("end", self.name, self.left < 0))
        return _StubPlanner.plan(self, begun)


def test_pipelined_event_loop_runs_whichever_scene_is_ready_and_waits_only_when_none_is():
    """PipelinedClosedLoops over planners with the three-piece surface (CPU, stub planners with scripted readiness): every scene's pieces
    run in order start -> mid -> end, every scene gets n plans at the simulator times it plans at alone, every scene is in flight early,
    a slow scene does not hold the others back, and no piece is collected before it is ready (the loop polls
    the pieces in flight when no scene has anything ready or startable)."""
    from mind_amd.pipelined import PipelinedClosedLoops
    log = []
    polls = {"s1": (2, 3), "s2": (0, 0), "s3": (40, 40)}            # s3's device work is slow
    sims = [ClosedLoopSim(SynthWorld(n_agents=3, n_lanes=2, n_segs=6, seed=k), _ThreePiecePlanner("s%d" % k, log, *polls["s%d" % k]), episode_plans=7)
            for k in (1, 2, 3)]
    pc = PipelinedClosedLoops(sims)
    steps = pc.run_plans(5) + pc.run_plans(6)
    assert [s.n_plans for s in sims] == [11, 11, 11] and steps == sum(s.n_steps for s in sims)
    alone = ClosedLoopSim(SynthWorld(n_agents=3, n_lanes=2, n_segs=6, seed=1), _StubPlanner(), episode_plans=7)
    alone.run_plans(11)
    assert sims[0].planner.plans == alone.planner.plans and sims[0].n_steps == alone.n_steps and np.array_equal(sims[0].state, alone.state)
    for name in polls:                                               # per scene: start, mid, end, start, mid, end ...
        mine = [e[0] for e in log if e[1] == name]
        assert mine == ["start", "mid", "end"] * 11
    assert [e[:2] for e in log[:3]] == [("start", "s1"), ("start", "s2"), ("mid", "s2")]       # ready work is collected before more is started
    assert ("start", "s3") in [e[:2] for e in log[:6]]                                             # ... and every scene is in flight early
    # the fast scenes finish plans while the slow one is still in its first pieces
    first_s3_end = next(i for i, e in enumerate(log) if e[:2] == ("end", "s3"))
    assert sum(1 for e in log[:first_s3_end] if e[:2] == ("end", "s2")) >= 2
    # no piece is collected before its device work has finished: when no scene is ready the loop polls the pieces in flight (it blocks on
    # the oldest one only after seconds without any becoming ready)
    assert not [e for e in log if e[0] in ("mid", "end") and not e[2]]


@pytest.mark.gpu
def test_closed_loop_with_real_planner():
    import os
    from mind_amd.planners.mind.planner import MINDPlanner
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pl = MINDPlanner(os.path.join(root, "mind_amd", "planners", "mind", "configs", "synthetic.json"))
    w = SynthWorld(n_agents=6, n_lanes=3, n_segs=8, seed=1)
    sim = ClosedLoopSim(w, pl)
    sim.run_until(4.0)
    steps = sim.run_plans(4)
    assert steps in (16, 20, 21) and sim.n_plans == 4
    scen, traj = sim.last_result
    assert len(scen) == 1 and len(traj) == 1 and np.all(np.isfinite(sim.state))
    # the ego stays near its lane over the 0.4 s driven
    d = np.abs(w.lane_y(1, sim.state[0]) - sim.state[1])
    assert d < 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["demo_1", "demo_3"])
def test_closed_loop_on_recorded_demo_scene(scene):
    """f3 end to end: a recorded AV2 demo scene (compact fixture) replayed through the closed-loop driver with
    the real planner (formula weights + scripted branching).  Agents appear and disappear in the recording, so the
    planner's track bookkeeping sees ragged histories; prune_merge on the device and on the host must still hand the
    same scenario trees to the contingency planner."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import WORKLOADS, make_closed_loop
    runs = {}
    for glue in (True, False):
        pl, sim, w = make_closed_loop(dict(WORKLOADS[scene]))
        pl.scen_tree_gen.device_glue = glue
        assert sim.sim_time >= 4.0 - 1e-9 and not sim.enabled
        steps = sim.run_plans(3)
        assert steps in (11, 15, 16) and np.all(np.isfinite(sim.state)) and np.all(np.isfinite(pl.ctrl))
        runs[glue] = (pl.scen_tree_gen.get_scenario_tree(), pl.timing["best_traj_idx"], np.array(sim.state), len(pl.agent_obs))
        # ego stays on its recorded lane: within 1.5 m of the target polyline after 0.3 s of closed-loop driving
        from mind_amd.planners.mind import utils as U
        d = U.get_distances_to_polyline(np.asarray(w.gt_tgt_lane, dtype=np.float64), sim.state[None, :2].astype(np.float64))
        assert float(np.ravel(d)[0]) < 1.5
        assert len(pl.agent_obs) <= w.n_agents and len(pl.agent_obs) >= 2
    (td, best_d, st_d, n_d), (th_, best_h, st_h, n_h) = runs[True], runs[False]
    assert n_d == n_h and len(td) == len(th_) and len(td) >= 1
    scale = float(np.abs(st_d[:2]).max())                        # float32 world coordinates: ulp grows with |x|
    tol = max(2e-4, 4 * np.spacing(np.float32(scale)))
    for a, b in zip(td, th_):
        assert list(a.nodes.keys()) == list(b.nodes.keys())
        for k in a.nodes:
            da, db = a.nodes[k].data, b.nodes[k].data
            assert np.allclose(da[0], db[0], rtol=1e-5)
            assert da[1].shape == db[1].shape and np.abs(da[1] - db[1]).max() < tol
    assert best_d == best_h and np.abs(st_d - st_h).max() < 5e-2


def test_episodes_restart_the_scene():
    """episode_plans = E: after E planning cycles the scene starts over (history rebuilt up to the enable time); the
    replayed steps are not counted, every episode repeats the same trigger schedule."""
    w = SynthWorld(n_agents=3, n_lanes=2, n_segs=6, seed=1)

    class P(_StubPlanner):
        def __init__(self):
            super().__init__()
            self.agent_obs = {"x": 1}
            self.times = []

        def plan(self, lcl):
            self.times.append(round(lcl.ego_agent.timestep * 0.1, 3))
            self.agent_obs["seen"] = True
            return True, np.array([0.5, 0.0]), None

    p = P()
    sim = ClosedLoopSim(w, p, episode_plans=3)
    assert p.agent_obs == {}                                   # a new episode starts from an empty observation history
    sim.run_until(4.0)
    steps = sim.run_plans(7)
    assert sim.n_plans == 7 and sim.n_episodes == 2
    assert p.times == [4.0, 4.1, 4.2] * 2 + [4.0]
    assert steps == 1 + 5 + 5 + 1 + 5 + 5 + 1                  # only the steps that lead to a plan inside an episode count
    assert sim.sim_time < 4.1 and "seen" in p.agent_obs
    free = ClosedLoopSim(w, P())                               # default: one open-ended episode
    free.run_until(4.0)
    free.run_plans(7)
    assert free.n_episodes == 0 and free.planner.times[-1] == 4.6
