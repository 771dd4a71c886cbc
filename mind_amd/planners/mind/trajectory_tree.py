"""Contingency planner: scenario tree -> cost tree -> tree-iLQR on the MI355X.

Call surface of the reference ``planners/mind/trajectory_tree.py:12-177``
(``init_warm_start_cost_tree`` / ``warm_start_solve`` / ``init_cost_tree`` / ``solve`` / ``debug``).
The reference builds a 256x256 float64 potential field per trajectory node on the CPU and runs a
Python iLQR; here ``init_*`` only flattens the scenario tree (LIFO DFS, every even sub-step -> one
trajectory node, Q13) and the solve is one persistent HIP kernel (libmind_hip.so:
mind_ilqr_solve_trees).  ``solve_batch`` solves all scenario trees of a plan in two launches.
"""
import os

import numpy as np

from ... import _lib
from ...runtime import get_runtime
from ..basic.tree import Node, Tree


def flatten_scenario_tree(scen_tree):
    """-> dict(parent i32 [M], prob f32 [M], mean f32 [M,a,2], cov f32 [M,a]); trajectory node keys are
    the array indices (creation order of trajectory_tree.py:30-54: stack pop() = last child first)."""
    parent, prob, mean, cov = [], [], [], []
    last = {}
    count = 0
    stack = [scen_tree.get_root()]
    while stack:
        node = stack.pop()
        p, trajs, covs = node.data[0], node.data[1], node.data[2]
        n = (trajs.shape[1] + 1) // 2                          # every even sub-step -> one trajectory node, chained
        if n:
            par = np.arange(count - 1, count + n - 1, dtype=np.int32)
            par[0] = last[node.parent_key] if node.parent_key is not None else -1
            parent.append(par)
            prob.append(np.full(n, p, np.float32))
            mean.append(np.transpose(trajs[:, ::2, :], (1, 0, 2)))
            cov.append(np.transpose(covs[:, ::2, 0], (1, 0)))
            count += n
            last[node.key] = count - 1
        else:
            last[node.key] = last[node.parent_key] if node.parent_key is not None else -1
        stack.extend(scen_tree.get_node(k) for k in node.children_keys)
    return dict(parent=np.concatenate(parent), prob=np.concatenate(prob),
                mean=np.ascontiguousarray(np.concatenate(mean), np.float32),
                cov=np.ascontiguousarray(np.concatenate(cov), np.float32))


_CFG_CACHE = {}


def ilqr_cfg_from(config, block, max_iter=100):
    """TrajTreeCfg (w_opt_cfg / opt_cfg dict) -> C struct (built once per config object and block: it is read-only afterwards)."""
    key = (id(config), block, max_iter)
    fp = _cfg_fingerprint(config, block)       # the reference re-reads the weights on every plan: an in-place edit must not be ignored
    hit = _CFG_CACHE.get(key)
    if hit is not None and hit[0] is config and hit[2] == fp:
        return hit[1]
    c = _ilqr_cfg_build(config, block, max_iter)
    _CFG_CACHE[key] = (config, c, fp)
    return c


def _cfg_fingerprint(config, block):
    o = getattr(config, block)
    return (config.dt,) + tuple(np.asarray(v, np.float64).tobytes() if isinstance(v, (list, tuple, np.ndarray)) else v for _, v in sorted(o.items()))


def _ilqr_cfg_build(config, block, max_iter):
    o = getattr(config, block)
    for name in ("w_des_state", "w_state_con", "w_ctrl"):
        m = np.asarray(o[name], np.float64)
        if m.ndim != 2 or np.any(m - np.diag(np.diag(m)) != 0.0):
            # the kernel carries the diagonals of the reference's weight matrices (all of its planning configs are
            # diagonal, configs/planning/demo_*.py:21-81); a full matrix would silently be a different problem
            raise ValueError(f"{block}.{name}: only diagonal weight matrices are supported by the tree-iLQR kernel")
    c = _lib.IlqrCfg()
    c.dt, c.wheelbase = config.dt, 2.5                       # trajectory_tree.py:15
    c.w_des_state[:] = list(np.diag(o["w_des_state"]))
    c.w_state_con[:] = list(np.diag(o["w_state_con"]))
    c.state_lower[:] = list(o["state_lower_bound"])
    c.state_upper[:] = list(o["state_upper_bound"])
    c.w_ctrl[:] = list(np.diag(o["w_ctrl"]))
    c.w_tgt = o["w_tgt"]
    c.w_ego, c.w_ego_cov_offset = o.get("w_ego", 0.0), o.get("w_ego_cov_offset", 0.0)
    c.w_exo, c.w_exo_cov_offset, c.w_exo_cost_offset = o.get("w_exo", 0.0), o.get("w_exo_cov_offset", 0.0), o.get("w_exo_cost_offset", 0.0)
    c.grid_res = o["smooth_grid_res"]
    c.grid_w, c.grid_h = o["smooth_grid_size"][1], o["smooth_grid_size"][0]
    c.max_iter = max_iter
    return c


class _LazyTrajTree(Tree):
    """Trajectory tree whose Node objects are only created when somebody looks at them (a plan builds one tree per scenario
    tree and keeps the cheapest: the others are only ever read through ``_arrays`` by evaluate_traj_trees)."""

    def __init__(self, parents, x0, xs, us, action_size):
        # no Tree.__init__: nodes / root / _leaves appear on first access (__getattr__)
        self._lazy = (parents, x0, xs, us, action_size)
        self.n_nodes = len(parents) + 1
        # node-order arrays for evaluate_traj_tree (root first), so that it does not rebuild them from the nodes
        self._arrays = (np.concatenate([np.asarray(x0, np.float64)[None], xs]), np.concatenate([np.zeros((1, action_size)), us]))

    def __getattr__(self, name):
        if name in ("nodes", "root", "_leaves") and "_lazy" in self.__dict__:
            par, x0, xs, us, action_size = self.__dict__.pop("_lazy")
            Tree.__init__(self)
            self.add_node(Node(-1, None, [x0, np.zeros(action_size)]))
            par = par.tolist()
            for k in range(len(par)):
                self.add_node(Node(k, par[k], [xs[k], us[k]]))
            return self.__dict__[name]
        raise AttributeError(name)

    def size(self):
        return self.n_nodes


def to_traj_tree(flat, x0, xs, us, action_size=2):
    """root key -1 holds [x0, 0]; node k holds [xs[k], us[k]] (trajectory_tree.py:140-146)."""
    return _LazyTrajTree(flat["parent"], x0, xs, us, action_size)


class _NoCall:
    """makes the side context exist and its first-call allocations happen (TrajectoryTreeOptimizer.prewarm): a three-node lane-only fit --
    device arena, page-locked staging, the kernel's attributes on that context -- whose result nobody reads"""

    def __init__(self, lib, cfg):
        self.lib, self.cfg = lib, cfg

    def run(self, rt):
        from ...predictor import IlqrCall
        flat = dict(parent=np.array([-1, 0, 1], np.int32), prob=np.ones(3, np.float32), mean=np.zeros((3, 1, 2), np.float32), cov=np.zeros((3, 1), np.float32))
        lane = np.array([[0.0, 0.0], [50.0, 0.0]])
        try:
            IlqrCall(self.lib, self.cfg, [flat], np.array([0.0, 0.0, 1.0, 0.0, 0.0, 0.0]), lane, 1.0, use_exo=0, background=True).run(rt)
        except Exception:      # noqa: BLE001
            pass
        return self


class _SideContext:
    """One background thread with its OWN HIP context on its own stream.  It only ever executes prepared tree-iLQR
    calls (`IlqrCall.run`: one C call, GIL released), so it neither competes for the interpreter with the main thread's
    AIME bookkeeping nor shares a stream with the predictor."""

    def __init__(self, device):
        from concurrent.futures import ThreadPoolExecutor
        self.device = device
        self.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mind-ilqr-side")
        self.rt = None

    def _run(self, call):
        if self.rt is None:
            import torch
            from ...predictor import HipPredictor
            self.stream = torch.cuda.Stream(self.device)
            with torch.cuda.stream(self.stream):
                self.rt = HipPredictor(self.device)          # bound to this stream
        return call.run(self.rt)

    def submit(self, call):
        return self.pool.submit(self._run, call)


class TrajectoryTreeOptimizer:
    def __init__(self, config=None, runtime=None):
        self.config = config
        self.rt = runtime
        self.shard = None       # mind_amd.parallel.Shard: deal the scenario trees round-robin over ranks
        self.solver = None      # injectable for tests: solver(cfg, flats, x0, lane, tv, use_exo, us_init) -> (xs, us, st)
        self.cost_tree = None
        self.debug = None
        self._job = None
        # speculative warm start (see speculate_warm)
        self.speculative = os.environ.get("MIND_SPECULATIVE_WARM_START", "1") != "0"
        self._worker = None
        self._last_structs = []
        self._spec = None
        self._spec_skip, self._spec_backoff, self._spec_last = 0, 4, None     # back-off doubles after every failed probe (4 .. 256 cycles)
        # contingency solves started ahead of solve_batch (solve_batch_begin): the kernel runs while the caller builds the scenario
        # trees' Python objects
        self.overlap = os.environ.get("MIND_ILQR_OVERLAP", "1") != "0"
        self._pending = None
        self.counters = {"solves": 0, "iterations": 0, "warm_speculated": 0, "warm_hits": 0}

    def _runtime(self):
        if self.rt is None:
            self.rt = get_runtime()
        return self.rt

    def _get_init_state(self, init_state, init_ctrl):
        return np.array([init_state[0], init_state[1], init_state[2], init_state[3], init_ctrl[0], init_ctrl[1]],
                        dtype=np.float64)

    def _prepare(self, scen_tree, init_state, init_ctrl, target_lane, target_vel, warm):
        self._job = dict(flat=flatten_scenario_tree(scen_tree), x0=self._get_init_state(init_state, init_ctrl),
                         lane=np.asarray(target_lane, np.float64), tv=float(target_vel), warm=warm)
        self.cost_tree = self._job["flat"]

    def init_warm_start_cost_tree(self, scen_tree, init_state, init_ctrl, target_lane, target_vel):
        self._prepare(scen_tree, init_state, init_ctrl, target_lane, target_vel, True)

    def init_cost_tree(self, scen_tree, init_state, init_ctrl, target_lane, target_vel):
        self._prepare(scen_tree, init_state, init_ctrl, target_lane, target_vel, False)

    def _solve(self, us_init):
        j = self._job
        cfg = ilqr_cfg_from(self.config, "w_opt_cfg" if j["warm"] else "opt_cfg")
        xs, us, st = self._runtime().ilqr_solve(cfg, [j["flat"]], j["x0"], j["lane"], j["tv"], 0 if j["warm"] else 1,
                                                None if us_init is None else [np.asarray(us_init, np.float64)])
        self.debug = st[0]
        return xs[0], us[0]

    def warm_start_solve(self, us_init=None):
        return self._solve(us_init)

    def solve(self, us_init=None):
        xs, us = self._solve(us_init)
        return to_traj_tree(self._job["flat"], self._job["x0"], xs, us, self.config.action_size)

    # ---- speculative warm start ---------------------------------------------------------------------------
    # The warm-start fit (planner.py:174-176, trajectory_tree.py:19-60) sees the cost tree's SHAPE (parents, node
    # probabilities), the ego state, the target lane and velocity -- no prediction.  All but the shape are known before
    # the AIME rounds start, and consecutive planning cycles usually grow trees of the same shape.  So the fits of the
    # previous cycle's shapes are started on a second stream when a cycle begins and run beside the predictor; after
    # AIME, every tree whose shape was guessed right takes its warm-start controls from there (bit-identical: same
    # kernel, same inputs) and only runs the full fit, the others run both fits as before.
    @staticmethod
    def _struct_key(par, prob):
        return (len(par), par.tobytes(), prob.tobytes())

    def speculate_warm(self, init_state, init_ctrl, target_lane, target_vel):
        if not self.speculative or self.solver is not None or self.shard is not None or not self._last_structs:
            return
        # a speculated fit only pays when the tree shape (parents + node probabilities) recurs: after a cycle in which
        # fewer than half of the guesses were used, skip the next `_spec_backoff` cycles (doubling after every failed probe), then probe again
        if self._spec_skip > 0:
            self._spec_skip -= 1
            self._spec_last = None
            return
        from ...predictor import IlqrCall
        x0 = self._get_init_state(init_state, init_ctrl)
        lane = np.array(target_lane, dtype=np.float64)
        # the warm-start cost has no agent term: one dummy agent per node
        flats = [dict(parent=par, prob=prob, mean=np.zeros((len(par), 1, 2), np.float32), cov=np.zeros((len(par), 1), np.float32))
                 for par, prob in self._last_structs]
        call = IlqrCall(self._runtime().lib, ilqr_cfg_from(self.config, "w_opt_cfg"), flats, x0, lane, target_vel, use_exo=0, background=True)
        self._spec = dict(x0=x0, lane=lane, tv=float(target_vel), structs=self._last_structs, call=call,
                          fut=self._side().submit(call))
        self.counters["warm_speculated"] += len(self._last_structs)
        self._spec_last = len(self._last_structs)

    def _side(self):
        if self._worker is None:
            self._worker = _SideContext(self._runtime().device)
        return self._worker

    def prewarm(self):
        """One-time set-up that would otherwise land in the middle of a closed loop: the side context of the speculative warm start (a thread, a HIP
        context on its own stream: 5 ms the first time a process creates one) is created at the FIRST plan instead of at the first probe a few
        cycles later (a fresh process ran its first 20 timed cycles 8 % slower for it, tests/diag/gpu_cold_start.py)."""
        if self.speculative and self.solver is None and self.shard is None and self._worker is None and not getattr(self, "_prewarmed", False):
            self._prewarmed = True
            try:
                self._side().pool.submit(self._side()._run, _NoCall(self._runtime().lib, ilqr_cfg_from(self.config, "w_opt_cfg"))).result()
            except Exception:      # noqa: BLE001      (no GPU / no library: the probe will report it)
                pass

    def _take_speculation(self, flats, x0, lane, target_vel):
        """-> {index into flats: (us_warm, stats_warm)} for the trees whose warm-start fit is already done."""
        spec, self._spec = self._spec, None
        if spec is None:
            return {}
        try:
            spec["fut"].result()
            _, us, st = spec["call"].finish()
        except _lib.MindError:                # the in-line path below then computes (or reports) the same thing
            self.counters["warm_failed"] = self.counters.get("warm_failed", 0) + len(spec["structs"])
            return {}
        if not (np.array_equal(spec["x0"], x0) and np.array_equal(spec["lane"], lane) and spec["tv"] == float(target_vel)):
            return {}
        have = {}
        for j, (par, prob) in enumerate(spec["structs"]):
            have.setdefault(self._struct_key(par, prob), j)
        hits = {}
        for i, f in enumerate(flats):
            j = have.get(self._struct_key(f["parent"], f["prob"]))
            if j is not None:
                hits[i] = (us[j], st[j])
        return hits

    def solve_batch_begin(self, flats, init_state, init_ctrl, target_lane, target_vel):
        """Start the contingency solves of a plan BEFORE its scenario trees exist as Python objects: `flats` are the flattened cost
        trees mind_aime_plan returned with the plan.  mind_ilqr_contingency_begin queues upload, launch (warm-start + full fit of every
        tree) and read-backs on this planner's own context and returns; the kernel runs while the caller builds the trees, and the next
        solve_batch() with the same arguments collects it (mind_ilqr_finish) instead of launching.  (A worker thread for the blocking
        call was measured first: 1 033 vs 1 083 sim steps/s -- the hand-over through the interpreter lock costs more than it hides.)  Only the plain case is taken ahead (no speculated warm start waiting, no shard,
        no injected solver); returns whether a launch was started."""
        self._drop_pending()
        if not self.overlap or self.solver is not None or self.shard is not None or self._spec is not None or not flats:
            return False
        from ...predictor import PlanIlqrCall
        rt = self._runtime()
        x0 = self._get_init_state(init_state, init_ctrl)
        lane = np.asarray(target_lane, np.float64)
        # (the flats are the library's own tables of the plan it just returned: the call names them instead of passing them back)
        call = PlanIlqrCall(rt.lib, ilqr_cfg_from(self.config, "w_opt_cfg"), ilqr_cfg_from(self.config, "opt_cfg"), [len(f["parent"]) for f in flats],
                            x0, lane, target_vel)
        call.begin(rt)            # returns with the kernel queued; a failed begin is reported by the collecting solve_batch (call.finish)
        if call.rc == _lib.MIND_ESTATE:      # this runtime is not the one that planned (no tables there): pass the trees themselves
            from ...predictor import IlqrCall
            call = IlqrCall(rt.lib, ilqr_cfg_from(self.config, "w_opt_cfg"), flats, x0, lane, target_vel, cfg_full=ilqr_cfg_from(self.config, "opt_cfg"))
            call.begin(rt)
        self._pending = dict(call=call, flats=flats, x0=x0, lane=lane, tv=float(target_vel))
        return True

    def plan_solve_args(self, init_state, init_ctrl, target_lane, target_vel):
        """What a native AIME plan needs to begin this cycle's contingency solves itself (mind_aime_plan_in.solve_*), or None when the
        solves are not the plain case (a speculated warm start waiting, a shard, an injected solver, overlap off)."""
        self._drop_pending()
        if not self.overlap or self.solver is not None or self.shard is not None or self._spec is not None:
            return None
        return (ilqr_cfg_from(self.config, "w_opt_cfg"), ilqr_cfg_from(self.config, "opt_cfg"), self._get_init_state(init_state, init_ctrl),
                np.asarray(target_lane, np.float64), float(target_vel))

    def adopt_begun_solves(self, flats, solve, rt):
        """the native plan on `rt` began the solves of `flats` with `solve` = plan_solve_args(...): the next solve_batch collects them"""
        from ...predictor import BegunPlanIlqrCall
        cw, cf, x0, lane, tv = solve
        call = BegunPlanIlqrCall(rt, cw, cf, [len(f["parent"]) for f in flats], x0, lane, tv)
        self._pending = dict(call=call, flats=flats, x0=x0, lane=lane, tv=float(tv))

    def _drop_pending(self):
        pend, self._pending = self._pending, None
        if pend is not None:
            pend["call"].wait()          # (a begun call must be finished before the context takes another one)

    def _take_pending(self, flats, x0, lane, target_vel):
        pend, self._pending = self._pending, None
        if pend is None:
            return None
        pend["call"].wait()
        same = (len(pend["flats"]) == len(flats) and np.array_equal(pend["x0"], x0) and np.array_equal(pend["lane"], lane) and pend["tv"] == float(target_vel)
                and all(pf is f or all(np.array_equal(pf[k], f[k]) for k in ("parent", "prob", "mean", "cov")) for pf, f in zip(pend["flats"], flats)))
        return pend["call"] if same else None

    # all scenario trees of one plan: 2 launches (warm start, full) instead of 2 x n_trees solves
    def solve_batch(self, scen_trees, init_state, init_ctrl, target_lane, target_vel):
        flats = [getattr(t, "_flat", None) or flatten_scenario_tree(t) for t in scen_trees]     # `_flat`: built by mind_aime_plan already
        x0 = self._get_init_state(init_state, init_ctrl)
        lane = np.asarray(target_lane, np.float64)
        solve = self.solver if self.solver is not None else self._runtime().ilqr_solve
        mine = list(range(len(flats))) if self.shard is None else self.shard.round_robin(len(flats))
        xs, us, st_w, st = [], [], [], []
        if mine:
            sub = [flats[i] for i in mine]
            if self.solver is None:      # warm start + full solve in one launch
                cfg_w, cfg_f = ilqr_cfg_from(self.config, "w_opt_cfg"), ilqr_cfg_from(self.config, "opt_cfg")
                hits = self._take_speculation(sub, x0, lane, target_vel) if self.shard is None else {}
                self._last_structs = [(np.ascontiguousarray(f["parent"], np.int32), np.ascontiguousarray(f["prob"], np.float32)) for f in sub]
                self.counters["warm_hits"] += len(hits)
                if self._spec_last is not None:
                    if 2 * len(hits) < self._spec_last:
                        # the shapes did not recur: a probe costs ~1.5 % of a branching cycle (measured on demo_1), so wait twice
                        # as long before the next one
                        self._spec_skip = self._spec_backoff
                        self._spec_backoff = min(2 * self._spec_backoff, 256)
                    else:
                        self._spec_backoff = 4
                self._spec_last = None
                xs, us, st_w, st = [None] * len(sub), [None] * len(sub), [None] * len(sub), [None] * len(sub)
                from ...predictor import IlqrCall
                hit_idx = sorted(hits)
                miss_idx = [i for i in range(len(sub)) if i not in hits]
                rt = self._runtime()
                miss_call = miss_fut = None
                ahead = self._take_pending(sub, x0, lane, target_vel) if self.shard is None else None
                if ahead is not None and not hit_idx:      # started by solve_batch_begin while the trees were being built
                    miss_call = ahead
                elif miss_idx:             # both fits, as without speculation
                    miss_call = IlqrCall(rt.lib, cfg_w, [sub[i] for i in miss_idx], x0, lane, target_vel, cfg_full=cfg_f)
                    if hit_idx:          # ... on the side context, beside the full fits of the guessed trees
                        miss_fut = self._side().submit(miss_call)
                    else:
                        miss_call.run(rt)
                if hit_idx:              # warm-start controls are there already: full fit only
                    hx, hu, hs = rt.ilqr_solve(cfg_f, [sub[i] for i in hit_idx], x0, lane, target_vel, 1, [hits[i][0] for i in hit_idx])
                    for k, i in enumerate(hit_idx):
                        xs[i], us[i], st_w[i], st[i] = hx[k], hu[k], hits[i][1], hs[k]
                if miss_idx:
                    if miss_fut is not None:
                        miss_fut.result()
                    mx, mu_, mw, mf = miss_call.finish()
                    for k, i in enumerate(miss_idx):
                        xs[i], us[i], st_w[i], st[i] = mx[k], mu_[k], mw[k], mf[k]
                self.counters["solves"] += 2 * len(sub)
                self.counters["iterations"] += sum(s_["iterations"] for s_ in st_w) + sum(s_["iterations"] for s_ in st)
                # node-iterations and agent-weighted cost evaluations (bench: algorithmic bytes of the sweep)
                self.counters["node_iterations"] = self.counters.get("node_iterations", 0) + sum(
                    len(f["parent"]) * (w_["iterations"] + f_["iterations"]) for f, w_, f_ in zip(sub, st_w, st))
                self.counters["node_iterations_exo"] = self.counters.get("node_iterations_exo", 0) + sum(
                    len(f["parent"]) * f_["iterations"] * f["mean"].shape[1] for f, f_ in zip(sub, st))
            else:
                _, us_w, st_w = solve(ilqr_cfg_from(self.config, "w_opt_cfg"), sub, x0, lane, target_vel, 0, None)
                xs, us, st = solve(ilqr_cfg_from(self.config, "opt_cfg"), sub, x0, lane, target_vel, 1, us_w)
        if self.shard is not None and self.shard.sharded:
            # one packed all-gather of the [M, 8] float64 (xs | us) rows of every rank's trees
            from ...parallel import gather_round_robin
            local = (np.concatenate([np.concatenate([np.asarray(x, np.float64), np.asarray(u, np.float64)], axis=1) for x, u in zip(xs, us)])
                     if mine else np.zeros((0, 8)))
            res = gather_round_robin(self.shard, [len(f["parent"]) for f in flats], local)
            xs, us = [np.ascontiguousarray(r[:, :6]) for r in res], [np.ascontiguousarray(r[:, 6:]) for r in res]
        self.debug = dict(warm=st_w, full=st)
        return [to_traj_tree(f, x0, x, u, self.config.action_size) for f, x, u in zip(flats, xs, us)]
