"""ClosedLoopSim's steps behind one native call (mind_loop_*, include/mind_hip.h; mind_amd/csrc/loop.hip).

The interpreter's share of a planning cycle on the recorded demo scene was 0.7 ms of 3.4 (simulator steps, observation windows, track
marshalling, the two C calls' argument building, result objects, candidate evaluation: profiles/r06j_host_time_demo_1.txt).  With a
`NativeLoop` the library keeps the simulator state and the observation windows itself and runs Simulator.run_sim's step (simulator.py:
51-107) -> MINDAgent.observe / plan (agent.py:317-331) -> MINDPlanner.plan (planner.py:66-145) -> kine_propagate without returning to
Python; the plan's scenario / trajectory trees become Python objects only when `ClosedLoopSim.last_result` is read.

What the library replays is tabulated HERE, once per scene, with the driver's own observation code (`ClosedLoopSim._exo_observation`,
`MINDPlanner.to_object_state`): the windows then hold the float64 values the Python steps would have put there, and every plan is the
same bits as the Python driver's (tests/test_gpu_native_loop.py).

A loop applies while the planner is the plain native case (HIP predictor, native AIME plan with the device-built root, plan-begun
contingency solves, native evaluation, no shard, no scripted modes, no injected solver).  The predicate is re-checked before every call;
when it stops holding -- a test flips `device_root`, a driver calls `step_begin` / `plan_start` itself -- the loop is handed back:
the windows are exported into `planner.agent_obs` and the simulator carries on with its Python steps from the same state.
"""
import ctypes as C
import os

import numpy as np

from . import _lib


_TRIG_CB = []


def _numpy_trig_callbacks():
    """(tan, sincos) C callbacks into numpy for the functions whose float64 results differ from the C library's on this host, None where they
    are the same routine (probed on 20 000 arguments; numpy 2.2 on an AVX-512 host: tan differs in 0.5 % of them, sin / cos in none)"""
    if not _TRIG_CB:
        import math
        rng = np.random.default_rng(0)
        x = rng.uniform(-0.8, 0.8, 20000)
        same_tan = all(float(a) == math.tan(float(b)) for a, b in zip(np.tan(x), x))
        y = rng.uniform(-7.0, 7.0, 20000)
        same_sc = all(float(a) == math.cos(float(b)) for a, b in zip(np.cos(y), y)) and all(float(a) == math.sin(float(b)) for a, b in zip(np.sin(y), y))
        f64, np_tan, np_sin, np_cos = np.float64, np.tan, np.sin, np.cos

        def tan_cb(v):
            return float(np_tan(f64(v)))

        def sincos_cb(v, out):
            v = f64(v)
            out[0], out[1] = float(np_sin(v)), float(np_cos(v))
        _TRIG_CB.append((None if same_tan else _lib.LOOP_TAN_FN(tan_cb), None if same_sc else _lib.LOOP_SINCOS_FN(sincos_cb)))
    return _TRIG_CB[0]


class NativeLoop:
    @staticmethod
    def why_not(sim):
        """None when the native loop applies to this simulator + planner + world, else the reason (a string)"""
        from .planners.mind.planner import MINDPlanner
        pl, w = sim.planner, sim.world
        if os.environ.get("MIND_NATIVE_LOOP", "1") == "0":
            return "MIND_NATIVE_LOOP=0"
        if type(pl) is not MINDPlanner:
            return "the planner is not a MINDPlanner"
        gen, opt, net = pl.scen_tree_gen, pl.traj_tree_opt, pl.network
        if gen.network is not net or type(net).__name__ != "ScenePredNet" or getattr(net, "rt", None) is None or not getattr(net, "_loaded", False):
            return "the generator's network is not the HIP predictor itself"
        if not (gen.native_aime and gen.device_glue and gen.device_select and gen.device_root) or gen.shard is not None or gen.ego_idx != 0 or gen.config is None:
            return "the native AIME plan with the device-built root is not selected"
        if gen.obs_len != 50 or pl.obs_len != 50 or not (2 <= gen.pred_len <= 60):
            return "horizons"
        if opt.solver is not None or opt.shard is not None or not opt.overlap or opt._runtime() is not net.rt:
            return "the contingency solves are not the plain case"
        if os.environ.get("MIND_PLAN_BEGINS_SOLVES", "1") == "0" or not getattr(pl, "_native_eval", True):
            return "plan-begun solves / native evaluation switched off"
        for name in ("agent_state", "object_type", "agent_ids", "n_agents", "target_lane", "target_lane_info", "target_velocity"):
            if not hasattr(w, name):
                return f"the world has no {name}"
        if np.asarray(w.agent_state(0, 0.0)).dtype not in (np.float32, np.float64):
            return "agent states are neither float32 nor float64"
        if sim.episode_plans is None and not hasattr(w, "max_step"):
            return "an open-ended world without a last step"
        lane = np.asarray(w.target_lane)
        if lane.dtype not in (np.float32, np.float64) or lane.ndim != 2 or lane.shape[1] != 2 or np.any(np.all(lane[1:] == lane[:-1], axis=1)):
            return "the target lane is not a float polyline without zero-length segments"
        return None

    def __init__(self, sim):
        from types import SimpleNamespace
        from .planners.mind import utils as U
        from .planners.mind.trajectory_tree import ilqr_cfg_from, _cfg_fingerprint
        self.sim, self.lib = sim, _lib.load()
        pl, w = sim.planner, sim.world
        gen, opt = pl.scen_tree_gen, pl.traj_tree_opt
        self.rt = pl.network.rt
        # ---- the scene's planner constants, built by the planner's own code
        lcl = SimpleNamespace(target_lane=w.target_lane, target_lane_info=w.target_lane_info, target_velocity=w.target_velocity)
        lane, info = pl.resample_target_lane(lcl)
        gen.set_target_lane(lane, info)
        if len(gen.target_lane) < 12:
            raise ValueError("target lane shorter than 12 points")
        st = U._static_lane_pieces(w, 15.0, 10)
        if st["num_lanes"] == 0:
            raise ValueError("no lanes")
        gen.n_lanes = int(st["num_lanes"])
        keep = self._keep = {}
        f32 = lambda x: np.ascontiguousarray(x, np.float32)
        keep["tl"], keep["ti"] = f32(gen.target_lane), f32(gen.target_lane_info)
        keep["lpts"], keep["lfl"] = np.ascontiguousarray(st["pts"], np.float64), np.ascontiguousarray(st["flags"], np.int32)
        keep["cw"], keep["cf"] = ilqr_cfg_from(opt.config, "w_opt_cfg"), ilqr_cfg_from(opt.config, "opt_cfg")
        keep["gt"] = np.ascontiguousarray(np.asarray(pl.gt_tgt_lane, np.float64))
        ev = np.asarray(w.target_lane)
        keep["ev"] = np.ascontiguousarray(ev)
        # what must stay as it is for the loop to remain this planner's plan (checked before every call)
        self._gt_obj, self._gen_cfg, self._opt_cfg, self._net = pl.gt_tgt_lane, gen.config, opt.config, pl.network
        self._scen_fp = (gen.config.tar_time_ahead, gen.config.tar_dist_thres, gen.config.max_depth, gen.pred_len)
        self._opt_fp = (_cfg_fingerprint(opt.config, "w_opt_cfg"), _cfg_fingerprint(opt.config, "opt_cfg"))
        self._fp = _cfg_fingerprint
        self._world_fp = (w.target_lane, w.target_lane_info, w.target_velocity)
        # ---- the replayed scene, one row per simulator step
        self._tabulate()
        d = _lib.LoopDesc()
        tb = self._tab
        n_steps, n_tracks = tb["ego"].shape[0], tb["valid"].shape[1]
        d.n_tracks, d.n_steps, d.clamp_last = n_tracks, n_steps, int(tb["clamp"])
        d.ego_state, d.exo_obs, d.exo_valid = tb["ego"].ctypes.data, tb["exo"].ctypes.data, tb["valid"].ctypes.data
        d.ego_state_is_f32, d.ego_obs, d.ego_trig32 = int(tb["f32"]), tb["ego_obs"].ctypes.data, tb["trig32"].ctypes.data
        # numpy's elementary functions where they are not the C library's (np.tan on AVX-512 hosts): the plant then calls back into numpy
        self._tan_cb, self._sincos_cb = _numpy_trig_callbacks()
        d.tan_fn = C.cast(self._tan_cb, C.c_void_p) if self._tan_cb is not None else None
        d.sincos_fn = C.cast(self._sincos_cb, C.c_void_p) if self._sincos_cb is not None else None
        d.timestep, d.type_slot = tb["ts"].ctypes.data, tb["slot"].ctypes.data
        d.sim_step, d.plan_step, d.enable_time = float(sim.SIM_STEP), float(sim.PLAN_STEP), float(sim.enable_time)
        d.wheelbase, d.max_speed, d.max_steer, d.max_acc, d.max_dec = float(sim.WB), float(sim.MAX_SPD), float(sim.MAX_STR), 6.0, -6.0
        d.n_lanes, d.lane_pts, d.lane_flags = int(st["num_lanes"]), keep["lpts"].ctypes.data, keep["lfl"].ctypes.data
        d.n_lane_pts, d.target_lane, d.target_lane_info = len(keep["tl"]), keep["tl"].ctypes.data, keep["ti"].ctypes.data
        cfg = gen.config
        d.time_ahead, d.min_vel, d.dist_thres = float(cfg.tar_time_ahead), 0.5, float(cfg.tar_dist_thres)
        d.max_depth, d.max_rounds, d.pred_len, d.prob_floor = int(cfg.max_depth), 16, int(gen.pred_len), 0.0
        d.cfg_warm, d.cfg_full = C.addressof(keep["cw"]), C.addressof(keep["cf"])
        d.solve_n_lane_pts, d.solve_lane, d.target_vel = len(keep["gt"]), keep["gt"].ctypes.data, float(w.target_velocity)
        d.eval_n_lane_pts, d.eval_lane_is_f32, d.eval_lane = len(keep["ev"]), int(keep["ev"].dtype == np.float32), keep["ev"].ctypes.data
        # the optimizer's speculative warm start inside the loop (a second context of the loop's own): opt-in, MIND_NATIVE_SPECULATE=1 --
        # a lone loop gains nothing from it (profiles/r06q_*), several loops sharing the device do
        self.speculative = bool(opt.speculative) and os.environ.get("MIND_NATIVE_SPECULATE", "0") == "1"
        d.speculative = int(self.speculative)
        h = C.c_void_p()
        rc = self.lib.mind_loop_create(self.rt.ctx, C.byref(d), C.byref(h))
        _lib.check(self.lib, self.rt.ctx, rc, "mind_loop_create")
        self.h = h
        self._ctx_value = self.rt.ctx.value
        self.out = _lib.LoopOut()
        self._out_ref = C.byref(self.out)
        self._result = None            # the last plan's [[scenario tree], [trajectory tree]] once somebody asked for it
        cn, ts = opt.counters, pl.timing_sum
        self._base = dict(plans=ts["plans"], aime_s=ts["aime_s"], ilqr_s=ts["ilqr_s"], total_s=ts["total_s"], n_expanded=gen.n_expanded, solves=cn["solves"],
                          iterations=cn["iterations"], node_iterations=cn.get("node_iterations", 0), node_iterations_exo=cn.get("node_iterations_exo", 0),
                          warm_speculated=cn.get("warm_speculated", 0), warm_hits=cn.get("warm_hits", 0))

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h is not None:
            self.lib.mind_loop_destroy(h)         # (host memory only: safe after the context is gone)

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass

    # ------------------------------------------------------------------------------------------
    def _tabulate(self):
        """ego_state / exo_obs / exo_valid / timestep per simulator step of an episode, from the simulator's own observation code.  Only the
        steps on which the planner is triggered read the exo rows: with a bounded episode the trigger steps are found by replaying the
        trigger arithmetic (the same float additions the library performs), the other rows stay empty."""
        sim = self.sim
        w, pl = sim.world, sim.planner
        n = int(w.n_agents)
        idx = {aid: i for i, aid in enumerate(w.agent_ids)}
        if len(idx) != n:
            raise ValueError("agent ids are not unique")
        if sim.episode_plans is not None:
            # replay check_enable / check_trigger (closed_loop.py step_begin) until the episode's last plan + one more cycle
            t, last, enabled, plans, trig = 0.0, None, False, 0, []
            k = 0
            while plans <= sim.episode_plans + 1 and k < 1000000:
                if t >= sim.enable_time:
                    enabled = True
                if last is None or (t - last) >= sim.PLAN_STEP:
                    last = t
                    trig.append(k)
                    plans += enabled
                t += sim.SIM_STEP
                k += 1
            n_tab, clamp, full = k, False, set(trig)
        else:
            n_tab, clamp, full = int(w.max_step) + 2, True, None
        ego = np.zeros((n_tab, 4))
        ego_obs = np.zeros((n_tab, 5))
        trig32 = np.zeros((n_tab, 2), np.float32)
        from types import SimpleNamespace
        exo = np.zeros((n_tab, n, 5))
        valid = np.zeros((n_tab, n), np.uint8)
        ts = np.zeros(n_tab, np.int32)
        t = 0.0
        for k in range(n_tab):
            es = w.agent_state(0, t)
            ego[k] = es
            ts[k] = int(round(t / 0.1))
            o = pl.to_object_state(SimpleNamespace(state=es, timestep=int(ts[k])))        # the recorded ego's window entry, by the driver's own code
            ego_obs[k] = (o.position[0], o.position[1], o.heading, o.velocity[0], o.velocity[1])
            if es.dtype == np.float32:
                trig32[k] = (np.cos(es[3]), np.sin(es[3]))          # numpy's float32 routines on the float32 yaw (kine_propagate's first step after the take-over)
            if full is None or k in full:
                for a in sim._exo_observation(t):
                    o = pl.to_object_state(a)
                    i = idx[a.id]
                    exo[k, i] = (o.position[0], o.position[1], o.heading, o.velocity[0], o.velocity[1])
                    valid[k, i] = 1
            t += sim.SIM_STEP
        from .planners.mind.utils import _TYPE_SLOT, _name
        slot = np.array([_TYPE_SLOT.get(_name(w.object_type(i)), 6) for i in range(n)], np.int32)
        self._tab = dict(ego=ego, ego_obs=ego_obs, trig32=trig32, exo=exo, valid=valid, ts=ts, slot=slot, clamp=clamp,
                         f32=w.agent_state(0, 0.0).dtype == np.float32)

    # ------------------------------------------------------------------------------------------
    def ok(self):
        """the planner is still the case this loop was built for (cheap: attribute reads and two small fingerprints)"""
        pl = self.sim.planner
        gen, opt, w = pl.scen_tree_gen, pl.traj_tree_opt, self.sim.world
        cfg = gen.config
        ctx = self.rt.ctx         # (a runtime that was closed, or re-created, under the loop: the library's loop holds the old context)
        return (self.h is not None and ctx is not None and ctx.value == self._ctx_value and gen.native_aime and gen.device_root and gen.device_glue and gen.device_select and gen.shard is None
                and gen.network is self._net and pl.network is self._net and opt.solver is None and opt.shard is None and opt.overlap
                and (not self.speculative or opt.speculative) and pl.gt_tgt_lane is self._gt_obj and cfg is self._gen_cfg and opt.config is self._opt_cfg and pl._native_eval
                and (cfg.tar_time_ahead, cfg.tar_dist_thres, cfg.max_depth, gen.pred_len) == self._scen_fp
                and w.target_lane is self._world_fp[0] and w.target_lane_info is self._world_fp[1] and w.target_velocity == self._world_fp[2]
                and (self._fp(self._opt_cfg, "w_opt_cfg"), self._fp(self._opt_cfg, "opt_cfg")) == self._opt_fp)

    def advance(self, until_plans=0, until_time=-1.0, max_steps=1):
        """mind_loop_advance + the simulator's / planner's mirrors of what happened; returns the number of plans computed"""
        sim, o = self.sim, self.out
        p0, s0 = o.n_plans, o.n_steps
        rc = self.lib.mind_loop_advance(self.h, int(until_plans), float(until_time), int(max_steps), self._out_ref)
        if rc != 0:
            msg = self.lib.mind_last_error_string(self.rt.ctx) or b""
            if rc == _lib.MIND_ESTATE and msg.startswith(b"unsupported"):
                return self._finish_step_on_the_host(p0, s0)
            _lib.check(self.lib, self.rt.ctx, rc, "mind_loop_advance")
        return self._mirror(p0, s0)

    def _mirror(self, p0, s0):
        sim, o = self.sim, self.out
        sim.sim_time, sim.enabled = o.sim_time, bool(o.enabled)
        sim.last_trigger = o.last_trigger if o.last_trigger >= 0.0 else None
        sim.state, sim.ctrl = np.array(o.state), np.array(o.ctrl)
        sim.n_steps += o.n_steps - s0
        dp = o.n_plans - p0
        if dp:
            sim.n_plans += dp
            self._result = None
            pl = sim.planner
            gen, opt = pl.scen_tree_gen, pl.traj_tree_opt
            nt = o.n_trees
            pl.timing = {"aime_s": o.aime_s, "ilqr_s": o.ilqr_s, "total_s": o.total_s, "nodes_expanded": o.n_expanded, "n_scen_trees": nt,
                         "best_traj_idx": o.best, "tree_costs": o.costs[:nt]}
            # the running totals are the library's (several plans may have run in this call): base values at the loop's creation + its sums
            t, b = o.tot, self._base
            ts, cn = pl.timing_sum, opt.counters
            ts["plans"], ts["aime_s"], ts["ilqr_s"], ts["total_s"] = b["plans"] + t.plans, b["aime_s"] + t.aime_s, b["ilqr_s"] + t.ilqr_s, b["total_s"] + t.total_s
            gen.n_expanded = b["n_expanded"] + t.expansions
            cn["solves"], cn["iterations"] = b["solves"] + 2 * t.scen_trees, b["iterations"] + t.iterations
            cn["node_iterations"], cn["node_iterations_exo"] = b["node_iterations"] + t.node_iterations, b["node_iterations_exo"] + t.node_iterations_exo
            cn["warm_speculated"], cn["warm_hits"] = b["warm_speculated"] + t.warm_speculated, b["warm_hits"] + t.warm_hits
            gen.n_native_plans += dp
            gen.branch_depth = o.n_rounds
        return dp

    def totals(self):
        """mind_loop_totals as a dict (running sums over the loop's plans; kernel durations only while profiling is on)"""
        t = self.out.tot
        d = {k: getattr(t, k) for k, _ in _lib.LoopTotals._fields_ if k != "ilqr_prof"}
        d["ilqr_prof"] = list(t.ilqr_prof)
        return d

    # ------------------------------------------------------------------------------------------
    def last_result(self):
        """[[scenario tree], [trajectory tree]] of the last plan (MINDPlanner.plan's third return value), built from the library's tables (raises
        MindError when another planner has planned on the shared context since: read the result before that, as a recorder does every step)"""
        if self._result is not None:
            return self._result
        if self.out.n_plans == 0:
            return None
        if self.rt.ctx is None or self.rt.ctx.value != self._ctx_value:
            raise _lib.MindError("the runtime of this loop was closed: its last plan can no longer be read")
        from .planners.mind.trajectory_tree import to_traj_tree
        pl = self.sim.planner
        gen, opt, w = pl.scen_tree_gen, pl.traj_tree_opt, self.sim.world
        po = _lib.AimePlanOut()
        ptr = [C.c_void_p() for _ in range(6)]
        x0 = np.zeros(6)
        rc = self.lib.mind_loop_last_plan(self.h, C.byref(po), *[C.byref(p) for p in ptr], x0.ctypes.data)
        _lib.check(self.lib, self.rt.ctx, rc, "mind_loop_last_plan")
        a, nt = self.out.n_agents, po.n_trees
        res = self.rt._aime_plan_result(0, po, a, int(self._keep["lfl"].shape[0]))
        tracks = np.frombuffer(C.string_at(ptr[4], a * 4), np.int32)
        types = np.frombuffer(C.string_at(ptr[5], a * 50 * 7 * 4), np.float32).reshape(a, 50, 7).astype(np.int16)
        root = {"TRAJS_TYPE": types, "TRAJS_TID": ["AV" if t == 0 else w.agent_ids[t] for t in tracks],
                "TRAJS_CAT": ["av" if i == 0 else "exo" for i in range(a)]}
        scen = gen._native_trees(res, root, None, count=False)
        off = np.frombuffer(C.string_at(po.tree_off, (nt + 1) * 4), np.int32)
        M = int(off[-1])
        xs = np.frombuffer(C.string_at(ptr[0], M * 48), np.float64).reshape(M, 6)
        us = np.frombuffer(C.string_at(ptr[1], M * 16), np.float64).reshape(M, 2)
        stats = lambda p: [dict(iterations=s.iterations, converged=s.converged, J=s.J, mu=s.mu)
                           for s in C.cast(p, C.POINTER(_lib.IlqrStats * nt)).contents]
        opt.debug = dict(warm=stats(ptr[2]), full=stats(ptr[3]))
        trajs = [to_traj_tree(t._flat, x0, xs[off[i]:off[i + 1]], us[off[i]:off[i + 1]], opt.config.action_size) for i, t in enumerate(scen)]
        self._all = (scen, trajs)
        b = self.out.best
        self._result = [[scen[b]], [trajs[b]]]
        return self._result

    # ------------------------------------------------------------------------------------------
    def hand_back(self):
        """the simulator continues with its Python steps: the windows go into planner.agent_obs (Track objects with their array mirrors,
        as MINDPlanner.update_observation keeps them), the simulator's fields are the loop's; the loop is closed"""
        from .planners.mind.planner import ObjectState, Track, TrackCategory
        sim = self.sim
        pl, w = sim.planner, sim.world
        self.lib.mind_loop_state(self.h, self._out_ref)
        o = self.out
        sim.sim_time, sim.enabled = o.sim_time, bool(o.enabled)
        sim.last_trigger = o.last_trigger if o.last_trigger >= 0.0 else None
        sim.state, sim.ctrl = np.array(o.state), np.array(o.ctrl)
        n_tracks = self._tab["valid"].shape[1]
        n = C.c_int(0)
        track, count = np.zeros(n_tracks, np.int32), np.zeros(n_tracks, np.int32)
        rows = np.zeros((n_tracks, 50, 7))
        rc = self.lib.mind_loop_export(self.h, n_tracks, C.byref(n), track.ctypes.data, count.ctypes.data, rows.ctypes.data)
        _lib.check(self.lib, self.rt.ctx, rc, "mind_loop_export")
        pl.agent_obs.clear()
        for s in range(n.value):
            ti, cn = int(track[s]), int(count[s])
            tid = "AV" if ti == 0 else w.agent_ids[ti]
            tr = Track(tid, [ObjectState(bool(r[0]), int(r[6]), (r[1], r[2]), r[3], (r[4], r[5])) for r in rows[s, :cn]], w.object_type(ti),
                       TrackCategory.FOCAL_TRACK if ti == 0 else TrackCategory.TRACK_FRAGMENT)
            try:
                buf = np.empty((4 * pl.obs_len, 6))
                buf[:cn] = rows[s, :cn, :6]
                tr._buf, tr._i, tr._n, tr._arr = buf, cn, cn, buf[0:cn]
            except AttributeError:
                pass
            pl.agent_obs[tid] = tr
        if sim.enabled:
            pl.update_state_ctrl(sim.state, sim.ctrl)
        ctx = self.rt.ctx
        if self._result is None and o.n_plans and getattr(sim, "_last_result", None) is None and ctx is not None and ctx.value == self._ctx_value:
            try:
                self.last_result()
            except _lib.MindError:
                pass
        sim._last_result = self._result
        sim._native = None
        self.close()

    def _finish_step_on_the_host(self, p0, s0):
        """the library left this step's plan to the round-by-round path (mind_aime_plan 'unsupported: ...'): its observation update is
        done; the plan and the rest of the step run on the host, and so does everything after it"""
        sim = self.sim
        o = self.out
        sim.n_steps += o.n_steps - s0
        sim.n_plans += o.n_plans - p0
        self.hand_back()
        lcl = sim._observation()
        sim.planner.update_state_ctrl(lcl.ego_agent.state, sim.ctrl)
        sim.step_end(sim.planner.plan(lcl))
        return int(o.n_plans - p0) + 1
