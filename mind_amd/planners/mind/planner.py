"""``MINDPlanner`` facade with the call surface ``agent.py`` uses (reference planners/mind/planner.py:
12-224): ``MINDPlanner(config_dir)``, ``update_target_lane``, ``update_state_ctrl``, ``plan(lcl_smp) ->
(ok, ctrl(a, delta), [[scenario_tree], [trajectory_tree]])``, ``update_observation``.

plan() = AIME scenario tree (batched predictor on the GPU) -> contingency tree-iLQR for every
scenario tree (all trees in one launch) -> min-cost tree -> first control.
"""
import json
import re
import os
from collections import namedtuple
from importlib import import_module

import numpy as np
import torch

from .scenario_tree import ScenarioTreeGenerator
from .trajectory_tree import TrajectoryTreeOptimizer

try:   # the real AV2 schema when it is installed next to the reference
    from av2.datasets.motion_forecasting.data_schema import ObjectState, Track, TrackCategory
except Exception:   # same field order as av2 0.2.1 (constructed positionally, planner.py:61-64,69-71)
    ObjectState = namedtuple("ObjectState", "observed timestep position heading velocity")

    class Track:
        def __init__(self, track_id, object_states, object_type, category):
            self.track_id, self.object_states, self.object_type, self.category = track_id, object_states, object_type, category

    class TrackCategory:
        TRACK_FRAGMENT, UNSCORED_TRACK, SCORED_TRACK, FOCAL_TRACK = range(4)


def _import_cfg(name):
    try:
        return import_module(name)
    except ImportError:
        return import_module("mind_amd." + name)


def contextlib_null():
    import contextlib
    return contextlib.nullcontext()


class MINDPlanner:
    def __init__(self, config_dir):
        self.obs_len = 50
        self.plan_len = 50
        self.agent_obs = {}
        self.state = None
        self.ctrl = None
        self.gt_tgt_lane = None
        self.last_ctrl_seq = []
        self.timing = {}
        self.timing_sum = {"plans": 0, "aime_s": 0.0, "ilqr_s": 0.0, "total_s": 0.0}   # running totals over all plans
        self._native_eval = os.environ.get("MIND_NATIVE_EVAL", "1") != "0"
        if isinstance(config_dir, dict):
            self.planner_cfg = config_dir
        else:
            with open(config_dir, "r") as f:
                self.planner_cfg = json.load(f)
        self.init_device()
        self.init_network()
        self.init_scen_tree_gen()
        self.init_traj_tree_opt()

    # ------------------------------------------------------------------------------------------
    def init_device(self):
        if not (self.planner_cfg.get("use_cuda", True) and torch.cuda.is_available()):
            raise RuntimeError("MINDPlanner (MI355X build) needs a GPU: there is no CPU execution path")
        self.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))

    def init_network(self):
        self.network_cfg = _import_cfg(self.planner_cfg["network_config"]).NetCfg()
        net_cfg = self.network_cfg.get_net_cfg()
        from .networks.network import ScenePredNet
        self.network = ScenePredNet(net_cfg, self.device, own_context=bool(self.planner_cfg.get("own_context", False)))
        path = self.planner_cfg["ckpt_path"]
        m = re.fullmatch(r"formula(?:_(\w+))?:(\d+)", str(path))     # "formula:<seed>" or "formula_<variant>:<seed>" (mind_amd/weights.py);
        if m is not None:                                            # anything else -- "formula_runs/x.tar" included -- is a checkpoint file
            from ...weights import formula_state_dict
            sd = formula_state_dict(int(m.group(2)), as_torch=True, variant=m.group(1))
        else:
            sd = torch.load(path, map_location="cpu")["state_dict"]
        self.network.load_state_dict(sd)
        self.network = self.network.to(self.device)
        self.network.eval()

    def init_scen_tree_gen(self):
        cfg = _import_cfg(self.planner_cfg["planning_config"]).ScenTreeCfg()
        self.scen_tree_gen = ScenarioTreeGenerator(self.device, self.network, self.obs_len, self.plan_len, cfg)

    def init_traj_tree_opt(self):
        cfg = _import_cfg(self.planner_cfg["planning_config"]).TrajTreeCfg()
        self.traj_tree_opt = TrajectoryTreeOptimizer(cfg, self.network.rt)

    # ------------------------------------------------------------------------------------------
    def enable_sharding(self, group=None):
        """Shard every AIME round's scenes and the contingency solves over the ranks of a
        torch.distributed process group (one process per GPU)."""
        from ...parallel import Shard
        sh = Shard(group)
        # the native plan (mind_aime_plan) distributes its rounds itself through the group's collectives (MIND_NATIVE_SHARD=0: the
        # round-by-round host path of rounds 1-3, kept for wrapped networks and as the comparison)
        if os.environ.get("MIND_NATIVE_SHARD", "1") != "0":
            sh.attach(self.network.rt)
        self.scen_tree_gen.shard = sh
        self.traj_tree_opt.shard = sh
        return sh

    def to_object_state(self, agent):
        s = agent.state
        return ObjectState(True, agent.timestep, (s[0], s[1]), s[3], (s[2] * np.cos(s[3]), s[2] * np.sin(s[3])))

    def update_observation(self, lcl_smp):
        """50-frame sliding Track per agent; agents missing from this frame get an unobserved dummy."""
        seen = {"AV"}
        ego = lcl_smp.ego_agent
        if "AV" not in self.agent_obs:
            self.agent_obs["AV"] = Track("AV", [], ego.type, TrackCategory.FOCAL_TRACK)
        self.agent_obs["AV"].object_states.append(self.to_object_state(ego))
        for agent in lcl_smp.exo_agents:
            if agent.id not in self.agent_obs:
                self.agent_obs[agent.id] = Track(agent.id, [], agent.type, TrackCategory.TRACK_FRAGMENT)
            # (a driver that builds the replayed agents ahead of time -- ClosedLoopSim's idle hook -- may attach their ObjectState)
            self.agent_obs[agent.id].object_states.append(getattr(agent, "obj_state", None) or self.to_object_state(agent))
            seen.add(agent.id)
        for tr in self.agent_obs.values():
            if tr.track_id not in seen:
                last = tr.object_states[-1]
                tr.object_states.append(ObjectState(False, last.timestep, last.position, last.heading, last.velocity))
            if len(tr.object_states) > self.obs_len:
                tr.object_states.pop(0)
            # array mirror of the track [n, (observed, x, y, heading, vx, vy)] used by the featuriser
            # (a window into a per-track buffer that is compacted every 3 x obs_len appends: no allocation / concatenation per frame)
            s = tr.object_states[-1]
            try:
                buf = getattr(tr, "_buf", None)
                if buf is None:
                    buf, tr._i, tr._n = np.empty((4 * self.obs_len, 6)), 0, 0
                    tr._buf = buf
                i = tr._i
                if i == len(buf):
                    keep = self.obs_len - 1
                    buf[:keep] = buf[i - keep:i]
                    i = keep
                buf[i, 0], buf[i, 1], buf[i, 2], buf[i, 3], buf[i, 4], buf[i, 5] = s.observed, s.position[0], s.position[1], s.heading, s.velocity[0], s.velocity[1]
                tr._i, tr._n = i + 1, min(tr._n + 1, self.obs_len)
                tr._arr = buf[i + 1 - tr._n:i + 1]
            except AttributeError:      # immutable Track type: the featuriser falls back to the object list
                pass

    def update_state_ctrl(self, state, ctrl):
        self.state = state
        self.ctrl = ctrl

    def update_target_lane(self, gt_tgt_lane):
        self.gt_tgt_lane = gt_tgt_lane

    # ------------------------------------------------------------------------------------------
    def plan(self, lcl_smp):
        return self.plan_end(self.plan_begin(lcl_smp))

    def _on_own_stream(self):
        """A planner with a context of its own (planner config "own_context", runtime.new_runtime) runs its kernels on that context's
        stream; whatever the plan does through torch (the round-by-round AIME path: index_select / pruning tensors / the predictor's
        output buffers) must be ordered on the same stream, not on the thread's current one."""
        ts = getattr(getattr(self.network, "rt", None), "torch_stream", None)      # (a wrapped / injected network has no runtime of its own)
        if ts is None:
            import contextlib
            return contextlib.nullcontext()
        return torch.cuda.stream(ts)

    def plan_begin(self, lcl_smp, idle_hook=None):
        with self._on_own_stream():
            return self._plan_begin(lcl_smp, idle_hook)

    def plan_end(self, begun):
        with self._on_own_stream():
            return self._plan_end(begun)

    def _plan_begin(self, lcl_smp, idle_hook=None):
        """First half of plan(): the AIME rounds, the start of the contingency solves (when the native plan hands over its flattened cost
        trees) and the scenario trees' Python objects.  A driver that plans several scenes from one thread -- each planner on its own
        context, planner config "own_context" -- calls the next scene's plan_begin before this scene's plan_end: the device then runs
        this scene's tree-iLQR beside the next scene's predictor."""
        import time
        t0 = time.perf_counter()
        self.scen_tree_gen.reset()
        lane, info = self.resample_target_lane(lcl_smp)
        self.scen_tree_gen.set_target_lane(lane, info)
        n0 = self.scen_tree_gen.n_expanded
        if hasattr(self.traj_tree_opt, "prewarm"):
            self.traj_tree_opt.prewarm()          # (first plan only: one-time set-up of the speculative warm start's side context)
        # warm-start fits of the previous cycle's tree shapes start now, beside the predictor (trajectory_tree.py)
        self.traj_tree_opt.speculate_warm(self.state, self.ctrl, self.gt_tgt_lane, lcl_smp.target_velocity)
        # the native AIME plan hands its flattened cost trees over before the scenario trees exist as Python objects: the contingency
        # solves start there, on a worker thread, and run while this thread builds the trees (solve_batch below collects them)
        opt = self.traj_tree_opt
        solve, ahead = self._solve_hooks(lcl_smp)
        scen_trees = self.scen_tree_gen.branch_aime(lcl_smp, self.agent_obs, on_flats=ahead, **({"solve": solve} if solve is not None else {}))
        # the device is busy with the contingency solves begun above: a caller's hook runs here (the closed-loop driver prefetches the next
        # replayed observation), the solves are collected afterwards
        hook = idle_hook if idle_hook is not None else getattr(self, "idle_hook", None)
        if hook is not None and getattr(opt, "_pending", None) is not None:
            hook()
        t1 = time.perf_counter()
        return (lcl_smp, scen_trees, t0, t1, n0)

    # ---- plan() in three pieces, none of which blocks on the device when its *_ready() says so: a driver that plans several scenes from
    #      one thread (mind_amd/pipelined.py) keeps every scene's native AIME plan (a thread of the library) and contingency solves (queued
    #      on the scene's context) in flight while it does another scene's host work
    def _solve_hooks(self, lcl_smp):
        """(solve, on_flats) for this cycle's AIME plan: `solve` = what the native plan needs to begin the contingency solves itself,
        behind its last kernel (mind_aime_plan_in.solve_*; None when they are not the plain case), `on_flats` = the callback that takes
        the plan's flattened cost trees as soon as the native call returns -- it adopts the solves the library began, or begins them."""
        opt = self.traj_tree_opt
        if not hasattr(opt, "solve_batch_begin"):
            return None, None
        rt = getattr(self.network, "rt", None)
        solve = opt.plan_solve_args(self.state, self.ctrl, self.gt_tgt_lane, lcl_smp.target_velocity) \
            if (rt is not None and hasattr(opt, "plan_solve_args") and os.environ.get("MIND_PLAN_BEGINS_SOLVES", "1") != "0") else None

        def ahead(flats, begun=False):
            if begun and solve is not None:
                return opt.adopt_begun_solves(flats, solve, rt)
            return opt.solve_batch_begin(flats, self.state, self.ctrl, self.gt_tgt_lane, lcl_smp.target_velocity)
        ahead.takes_begun = True
        return solve, ahead

    def _torch_free(self):
        """this planner's pieces issue no torch operation (native AIME plan, native solves and evaluation): they need not switch torch's
        current stream to the context's -- torch.cuda.stream() costs 40 us a time, six times a cycle"""
        ok = getattr(self.scen_tree_gen, "_native_ok", None)
        # (a scripted network -- mind_amd/synth.py -- builds and uploads its mode tensors through torch before the native plan reads them on
        # the context's stream: those planners keep the stream switch)
        scripted = hasattr(getattr(self.scen_tree_gen, "network", None), "_modes")
        return contextlib_null() if (ok is not None and ok() and not scripted and self.traj_tree_opt.solver is None) else self._on_own_stream()

    def plan_start(self, lcl_smp):
        """host work before the AIME rounds + the start of the native plan; returns a token for plan_started_ready / plan_begin_finish"""
        with self._torch_free():
            import time
            t0 = time.perf_counter()
            self.scen_tree_gen.reset()
            lane, info = self.resample_target_lane(lcl_smp)
            self.scen_tree_gen.set_target_lane(lane, info)
            n0 = self.scen_tree_gen.n_expanded
            self.traj_tree_opt.speculate_warm(self.state, self.ctrl, self.gt_tgt_lane, lcl_smp.target_velocity)
            begin = getattr(self.scen_tree_gen, "branch_aime_begin", None)
            solve, ahead = self._solve_hooks(lcl_smp)
            tok = begin(lcl_smp, self.agent_obs, **({"solve": solve} if solve is not None else {})) if begin is not None else None
            return (lcl_smp, tok, t0, n0, solve, ahead)

    def plan_started_ready(self, started):
        """the native plan begun by plan_start has finished (plan_begin_finish will not wait for it)"""
        tok = started[1]
        return True if tok is None else self.scen_tree_gen.branch_aime_ready(tok)

    def plan_begin_finish(self, started):
        """collects the native plan, starts the contingency solves; returns what plan_begin returns"""
        import time
        lcl_smp, tok, t0, n0, solve, ahead = started
        if tok is None:
            with self._on_own_stream():
                scen_trees = self.scen_tree_gen.branch_aime(lcl_smp, self.agent_obs, on_flats=ahead, **({"solve": solve} if solve is not None else {}))
        else:       # (a plan the library hands back to the round-by-round path runs that path, torch operations included, on the context's stream)
            scen_trees = self.scen_tree_gen.branch_aime_finish(tok, on_flats=ahead, host_context=self._on_own_stream)
        return (lcl_smp, scen_trees, t0, time.perf_counter(), n0)

    def plan_end_piece(self, begun):
        """plan_end for the three-piece driver (no torch stream switch when the planner issues no torch operation)"""
        with self._torch_free():
            return self._plan_end(begun)

    def plan_end_ready(self, begun):
        """the contingency solves begun by plan_begin / plan_begin_finish have finished (plan_end will not wait for them)"""
        rt = getattr(self.network, "rt", None)
        return True if rt is None or not hasattr(rt, "busy") else not rt.busy()

    def _plan_end(self, begun):
        """Second half of plan(): collects (or runs) the contingency solves, evaluates the candidates, returns plan()'s result."""
        lcl_smp, scen_trees, t0, t1, n0 = begun
        if len(scen_trees) < 0:
            return False, None, None
        import time
        # (a pipelined driver runs other scenes' plan_begin between the two halves: their time is not this plan's)
        return self._solve_and_select(lcl_smp, scen_trees, t0, t1, n0, t_collect=time.perf_counter())

    def _solve_and_select(self, lcl_smp, scen_trees, t0, t1, n0, t_collect=None):
        """planner.py:120-145: contingency solve of every scenario tree, evaluation, the reference's strict `<` scan for the
        cheapest tree (the first minimum; a NaN cost never wins), first control.  Shared by plan() and plan_rounds()."""
        import time
        traj_trees = self.traj_tree_opt.solve_batch(scen_trees, self.state, self.ctrl, self.gt_tgt_lane,
                                                    lcl_smp.target_velocity)
        t2 = time.perf_counter()
        gap = 0.0 if t_collect is None else max(t_collect - t1, 0.0)      # time between plan_begin's return and plan_end's entry
        best, min_cost = None, np.inf
        costs = self.evaluate_traj_trees(lcl_smp, traj_trees)
        for i, cost in enumerate(costs):
            if cost < min_cost:
                min_cost, best = cost, i
        opt = traj_trees[best]
        nxt = opt.get_node(opt.get_root().children_keys[0])
        ret_ctrl = nxt.data[0][-2:]          # (a, delta) of the first rolled-out state (Q15)
        self.timing = {"aime_s": t1 - t0, "ilqr_s": t2 - t1 - gap, "total_s": time.perf_counter() - t0 - gap,
                       "nodes_expanded": self.scen_tree_gen.n_expanded - n0, "n_scen_trees": len(scen_trees),
                       "best_traj_idx": best, "tree_costs": [float(c) for c in costs]}
        self._accumulate_timing()
        return True, ret_ctrl, [[scen_trees[best]], [traj_trees[best]]]

    def _accumulate_timing(self):
        ts = self.timing_sum
        ts["plans"] += 1
        for k in ("aime_s", "ilqr_s", "total_s"):
            ts[k] += self.timing[k]

    def plan_rounds(self, lcl_smp):
        """plan() as a generator over the AIME rounds (ScenarioTreeGenerator.branch_aime_rounds): yields each round's
        (scenes, predictor inputs), is sent the predictor outputs, returns plan()'s result.  mind_amd.fused drives several
        planners' generators in lock-step and answers all their rounds with ONE predictor batch (BASELINE config 3)."""
        import time
        t0 = time.perf_counter()
        self.scen_tree_gen.reset()
        lane, info = self.resample_target_lane(lcl_smp)
        self.scen_tree_gen.set_target_lane(lane, info)
        n0 = self.scen_tree_gen.n_expanded
        scen_trees = yield from self.scen_tree_gen.branch_aime_rounds(lcl_smp, self.agent_obs)
        t1 = time.perf_counter()
        return self._solve_and_select(lcl_smp, scen_trees, t0, t1, n0)

    def resample_target_lane(self, lcl_smp):
        """1 m resampling of the target lane and its per-point info (planner.py:147-171).  A replayed scene hands over the SAME lane /
        info objects every cycle: the result is kept while they are (the arrays are only read downstream)."""
        key = getattr(self, "_rtl_key", None)
        fp = self._lane_fingerprint(lcl_smp)
        if key is not None and key[0] is lcl_smp.target_lane and key[1] is lcl_smp.target_lane_info and key[2] == fp:
            return self._rtl_val
        self._rtl_key, self._rtl_val = (lcl_smp.target_lane, lcl_smp.target_lane_info, fp), self._resample_target_lane(lcl_smp)
        return self._rtl_val

    @staticmethod
    def _lane_fingerprint(lcl_smp):
        """content hash of the target lane and its info arrays (a few KB): an in-place edit of the same objects is a new lane"""
        try:
            return hash((np.asarray(lcl_smp.target_lane).tobytes(),) + tuple(np.asarray(i).tobytes() for i in lcl_smp.target_lane_info))
        except Exception:
            return None

    def _resample_target_lane(self, lcl_smp):
        lane = np.asarray(lcl_smp.target_lane)
        infos = lcl_smp.target_lane_info
        seg = lane[1:] - lane[:-1]
        n = np.ceil(np.linalg.norm(seg, axis=1) / 1.0).astype(int)
        idx = np.repeat(np.arange(len(n)), n)
        j = np.arange(len(idx)) - np.repeat(np.cumsum(n) - n, n)
        alpha = (j / n[idx]).astype(lane.dtype if lane.dtype == np.float32 else np.float64)
        pts = lane[idx] + alpha[:, None] * seg[idx]
        pts = np.concatenate([pts, lane[-1:]], axis=0)
        idx = np.concatenate([idx, [len(lane) - 1]])
        return pts, [np.array(np.asarray(info)[idx]) for info in infos]

    def get_traj_tree(self, scen_tree, lcl_smp):
        o = self.traj_tree_opt
        o.init_warm_start_cost_tree(scen_tree, self.state, self.ctrl, self.gt_tgt_lane, lcl_smp.target_velocity)
        xs, us = o.warm_start_solve()
        o.init_cost_tree(scen_tree, self.state, self.ctrl, self.gt_tgt_lane, lcl_smp.target_velocity)
        return o.solve(us), o.debug

    def evaluate_traj_trees(self, lcl_smp, traj_trees):
        """evaluate_traj_tree for all candidate trees of a plan in one set of array ops (the per-tree sums are taken
        over each tree's own node range); falls back to the per-tree function for trees without solver arrays."""
        packs = [getattr(t, "_arrays", None) for t in traj_trees]
        # (a single candidate takes the same path: its cost decides nothing, and the native loop -- csrc/loop.hip -- prices every plan this way)
        if len(traj_trees) < 1 or any(p is None or len(p[0]) != t.size() for p, t in zip(packs, traj_trees)):
            return [self.evaluate_traj_tree(lcl_smp, t) for t in traj_trees]
        st = np.concatenate([p[0] for p in packs])
        ct = np.concatenate([p[1] for p in packs])
        counts = np.array([len(p[0]) for p in packs])
        lane = np.asarray(lcl_smp.target_lane)
        if getattr(self, "_native_eval", True) and lane.dtype in (np.float32, np.float64) and lane.ndim == 2 and st.dtype == np.float64 and ct.dtype == np.float64 \
                and not np.any(np.all(lane[1:] == lane[:-1], axis=1)):      # (a zero-length segment: the numpy path raises the reference's assertion)
            # the same arithmetic in native code (mind_eval_traj_trees: numpy's operation and summation order)
            import ctypes as C
            from ... import _lib
            lib = _lib.load()
            st, ct, lane = np.ascontiguousarray(st), np.ascontiguousarray(ct), np.ascontiguousarray(lane)
            cnt = np.ascontiguousarray(counts, np.int32)
            out = np.zeros(len(packs))
            dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
            rc = lib.mind_eval_traj_trees(dp(st), dp(ct), cnt.ctypes.data_as(C.POINTER(C.c_int32)), len(packs), C.c_void_p(lane.ctypes.data),
                                          int(lane.dtype == np.float32), len(lane), C.c_double(float(lcl_smp.target_velocity)), dp(out))
            if rc == 0:
                return list(out)
        starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
        lane = np.asarray(lcl_smp.target_lane)
        # x / y kept as separate [nodes, segments] arrays (same arithmetic per element as the [.., 2] form, a
        # fraction of the numpy overhead)
        sx, sy = lane[:-1, 0][None], lane[:-1, 1][None]
        dx, dy = lane[1:, 0][None] - sx, lane[1:, 1][None] - sy
        l2 = dx ** 2 + dy ** 2
        assert np.all(l2 != 0.0), "Polyline segments should not have zero lengths."
        px, py = st[:, 0][:, None], st[:, 1][:, None]
        t = np.clip(((px - sx) * dx + (py - sy) * dy) / l2, 0, 1)
        dist = np.sqrt((px - (sx + t * dx)) ** 2 + (py - (sy + t * dy)) ** 2).min(axis=1)
        per_node = (0.1 * ct[:, 0] ** 2 + 5.0 * ct[:, 1] ** 2) + 0.01 * (lcl_smp.target_velocity - st[:, 2]) ** 2 + 0.01 * dist
        return list(np.add.reduceat(per_node, starts) / counts)

    def evaluate_traj_tree(self, lcl_smp, traj_tree):
        """mean over nodes of .1 jerk^2 + 5 steer_rate^2 + .01 (v_tgt - v)^2 + .01 dist(target lane) (:180-198),
        all nodes at once."""
        nodes = list(traj_tree.nodes.values())
        packed = getattr(traj_tree, "_arrays", None)        # (states, controls) the solver already holds as arrays
        if packed is not None and len(packed[0]) == len(nodes):
            st, ct = packed
        else:
            st = np.array([n.data[0] for n in nodes], dtype=np.float64)
            ct = np.array([n.data[1] for n in nodes], dtype=np.float64)
        comfort = (0.1 * ct[:, 0] ** 2 + 5.0 * ct[:, 1] ** 2).sum()
        eff = (0.01 * (lcl_smp.target_velocity - st[:, 2]) ** 2).sum()
        lane = np.asarray(lcl_smp.target_lane)
        s, e = lane[:-1], lane[1:]
        d = e - s
        l2 = (d ** 2).sum(-1)
        assert np.all(l2 != 0.0), "Polyline segments should not have zero lengths."
        rel = st[:, None, :2] - s[None]
        t = np.clip((rel * d[None]).sum(-1) / l2[None], 0, 1)
        near = s[None] + t[..., None] * d[None]
        dist = np.sqrt(((st[:, None, :2] - near) ** 2).sum(-1)).min(axis=1)
        tgt = (0.01 * dist).sum()
        return (comfort + eff + tgt) / len(nodes)

    def get_dist_to_target_lane(self, lcl_smp, state):
        lane = lcl_smp.target_lane
        px, py = state[:2]
        sx, sy = lane[:-1].T
        ex, ey = lane[1:].T
        dx, dy = ex - sx, ey - sy
        l2 = dx ** 2 + dy ** 2
        assert np.all(l2 != 0.0), "Polyline segments should not have zero lengths."
        t = np.clip(((px - sx) * dx + (py - sy) * dy) / l2, 0, 1)
        nx, ny = sx + t * dx, sy + t * dy
        k = np.argmin(np.sqrt((px - nx) ** 2 + (py - ny) ** 2))
        return np.linalg.norm(np.array((nx[k], ny[k])) - state[:2])
