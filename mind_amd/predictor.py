"""Host-side handle on the HIP scene predictor (one context per process/GPU)."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib


# record layout of mind_aime_node (include/mind_hip.h)
AIME_NODE_DTYPE = np.dtype({"names": ["round", "scene", "mode", "parent", "prob", "cur_t", "end_t", "flags", "dur", "row_off", "tgt_pts"],
                            "formats": ["<i4", "<i4", "<i4", "<i4", "<f4", "<i4", "<i4", "<i4", "<i4", "<i8", ("<f4", (22,))],
                            "offsets": [0, 4, 8, 12, 16, 20, 24, 28, 32, 40, 48], "itemsize": 136})


class IlqrCall:
    """One tree-iLQR call split into its three parts, so that a caller can prepare the arguments on its own thread, hand
    only ``run`` (a single C call, which releases the GIL) to another thread / context and read the results later.
    cfg_full given: mind_ilqr_contingency (warm-start fit with ``cfg``, then full fit with ``cfg_full``); otherwise
    mind_ilqr_solve_trees with ``use_exo`` / ``us_init``."""

    def __init__(self, lib, cfg, flats, x0, lane, target_vel, cfg_full=None, use_exo=1, us_init=None, background=False):
        self.lib, self.cfg, self.cfg_full, self.use_exo = lib, cfg, cfg_full, int(use_exo)
        # a fit that runs beside the predictor (speculative warm start) stays on one workgroup per tree: the multi-workgroup
        # kernel's barrier spins would hold CUs the predictor needs
        self.background = bool(background)
        n = self.n = len(flats)
        self.trees = (_lib.CostTree * n)()
        self.keep, self.Ms = [], []
        for i, f in enumerate(flats):
            par = np.ascontiguousarray(f["parent"], np.int32)
            prob = np.ascontiguousarray(f["prob"], np.float32)
            mean = np.ascontiguousarray(f["mean"], np.float32)
            cov = np.ascontiguousarray(f["cov"], np.float32)
            self.keep += [par, prob, mean, cov]
            t = self.trees[i]
            t.n_nodes = len(par)
            t.parent = par.ctypes.data_as(C.POINTER(C.c_int32))
            t.prob = prob.ctypes.data_as(C.POINTER(C.c_float))
            t.n_agents = mean.shape[1]
            t.agent_mean = mean.ctypes.data_as(C.POINTER(C.c_float))
            t.agent_cov = cov.ctypes.data_as(C.POINTER(C.c_float))
            self.Ms.append(len(par))
        Mt = int(sum(self.Ms))
        self.x0 = np.ascontiguousarray(x0, np.float64)
        self.lane = np.ascontiguousarray(lane, np.float64)
        self.tv = C.c_double(float(target_vel))
        self.xs, self.us = np.zeros((Mt, 6)), np.zeros((Mt, 2))
        self.st = (_lib.IlqrStats * n)()
        self.st_full = (_lib.IlqrStats * n)() if cfg_full is not None else None
        self.ui = None if us_init is None else np.ascontiguousarray(np.concatenate(us_init), np.float64)
        self.rc, self.ctx = None, None

    def run(self, rt):
        """the launch (+ its synchronisation) on ``rt``'s context; nothing else happens here"""
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
        self.ctx = rt.ctx
        now = getattr(rt, "_ilqr_wgs_now", None)
        if self.background and now != 1:
            self.lib.mind_set_tuning(rt.ctx, b"ilqr_wgs", 1)
            rt._ilqr_wgs_now = 1
        elif not self.background and now == 1:          # back to what the user (or the library default) asked for
            rt._ilqr_wgs_now = getattr(rt, "_ilqr_wgs_user", int(os.environ.get("MIND_ILQR_WGS", "16")))
            self.lib.mind_set_tuning(rt.ctx, b"ilqr_wgs", rt._ilqr_wgs_now)
        if self.cfg_full is not None:
            self.rc = self.lib.mind_ilqr_contingency(rt.ctx, C.byref(self.cfg), C.byref(self.cfg_full), self.trees, self.n, dp(self.x0),
                                                     dp(self.lane), len(self.lane), self.tv, dp(self.xs), dp(self.us), self.st, self.st_full)
        else:
            self.rc = self.lib.mind_ilqr_solve_trees(rt.ctx, C.byref(self.cfg), self.trees, self.n, dp(self.x0), dp(self.lane),
                                                     len(self.lane), self.tv, self.use_exo, dp(self.ui), dp(self.xs), dp(self.us), self.st)
        return self

    def begin(self, rt):
        """upload + launch + queued read-backs of a contingency call (mind_ilqr_contingency_begin): returns without waiting for the
        kernel; ``wait`` collects it.  The caller's thread is free in between (the planner builds its Python trees there)."""
        assert self.cfg_full is not None and not self.background
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
        self.ctx = rt.ctx
        self._rt = rt
        if getattr(rt, "_ilqr_wgs_now", None) == 1:
            rt._ilqr_wgs_now = getattr(rt, "_ilqr_wgs_user", int(os.environ.get("MIND_ILQR_WGS", "16")))
            self.lib.mind_set_tuning(rt.ctx, b"ilqr_wgs", rt._ilqr_wgs_now)
        self.rc = self.lib.mind_ilqr_contingency_begin(rt.ctx, C.byref(self.cfg), C.byref(self.cfg_full), self.trees, self.n, dp(self.x0),
                                                       dp(self.lane), len(self.lane), self.tv, dp(self.xs), dp(self.us), self.st, self.st_full)
        self._begun = self.rc == 0
        return self

    def wait(self):
        if getattr(self, "_begun", False):
            self._begun = False
            rt = getattr(self, "_rt", None)
            if rt is not None and (rt.ctx is None or rt.ctx.value != self.ctx.value):
                # the runtime was closed (mind_ctx_destroy drained the stream and dropped the pending half): nothing to collect
                self.rc = _lib.MIND_ESTATE
            else:
                self.rc = self.lib.mind_ilqr_finish(self.ctx)
        return self

    def finish(self):
        name = "mind_ilqr_contingency" if self.cfg_full is not None else "mind_ilqr_solve_trees"
        _lib.check(self.lib, self.ctx, self.rc, name)
        offs = np.cumsum([0] + self.Ms)
        n = self.n
        stats = lambda s_: [dict(iterations=s_[i].iterations, converged=s_[i].converged, J=s_[i].J, mu=s_[i].mu) for i in range(n)]
        xs = [self.xs[offs[i]:offs[i + 1]] for i in range(n)]
        us = [self.us[offs[i]:offs[i + 1]] for i in range(n)]
        if self.cfg_full is not None:
            return xs, us, stats(self.st), stats(self.st_full)
        return xs, us, stats(self.st)


class PlanIlqrCall(IlqrCall):
    """The contingency call on the cost trees the context's last mind_aime_plan flattened (mind_ilqr_contingency_begin_plan): the tree
    arrays never leave the library, only the node counts are needed here to size and split the results."""

    def __init__(self, lib, cfg, cfg_full, node_counts, x0, lane, target_vel):
        self.lib, self.cfg, self.cfg_full, self.use_exo, self.background = lib, cfg, cfg_full, 1, False
        self.Ms = [int(m) for m in node_counts]
        n = self.n = len(self.Ms)
        Mt = int(sum(self.Ms))
        self.x0 = np.ascontiguousarray(x0, np.float64)
        self.lane = np.ascontiguousarray(lane, np.float64)
        self.tv = float(target_vel)
        self.xs, self.us = np.zeros((Mt, 6)), np.zeros((Mt, 2))
        self.st, self.st_full = (_lib.IlqrStats * n)(), (_lib.IlqrStats * n)()
        self.ui, self.rc, self.ctx = None, None, None

    def begin(self, rt):
        self.ctx = rt.ctx
        self._rt = rt              # (wait() refuses to collect from a runtime that was closed meanwhile)
        if getattr(rt, "_ilqr_wgs_now", None) == 1:
            rt._ilqr_wgs_now = getattr(rt, "_ilqr_wgs_user", int(os.environ.get("MIND_ILQR_WGS", "16")))
            self.lib.mind_set_tuning(rt.ctx, b"ilqr_wgs", rt._ilqr_wgs_now)
        self.rc = self.lib.mind_ilqr_contingency_begin_plan(rt.ctx, C.byref(self.cfg), C.byref(self.cfg_full), self.x0.ctypes.data, self.lane.ctypes.data,
                                                            len(self.lane), self.tv, self.xs.ctypes.data, self.us.ctypes.data, self.st, self.st_full)
        self._begun = self.rc == 0
        return self

    def run(self, rt):
        return self.begin(rt).wait()


class BegunPlanIlqrCall(PlanIlqrCall):
    """The contingency call mind_aime_plan began itself behind its plan (mind_aime_plan_in.solve_*, out.solves_begun): nothing to launch
    here, ``wait`` collects it (mind_ilqr_finish_plan copies the results out of the library)."""

    def __init__(self, rt, cfg, cfg_full, node_counts, x0, lane, target_vel):
        super().__init__(rt.lib, cfg, cfg_full, node_counts, x0, lane, target_vel)
        self.ctx, self._rt, self._begun, self.rc = rt.ctx, rt, True, 0

    def begin(self, rt):
        return self

    def wait(self):
        if getattr(self, "_begun", False):
            self._begun = False
            rt = self._rt
            if rt.ctx is None or rt.ctx.value != self.ctx.value:
                self.rc = _lib.MIND_ESTATE
            else:
                self.rc = self.lib.mind_ilqr_finish_plan(self.ctx, int(sum(self.Ms)), self.n, self.xs.ctypes.data, self.us.ctypes.data,
                                                         C.cast(self.st, C.c_void_p), C.cast(self.st_full, C.c_void_p))
        return self


class HipPredictor:
    """Owns a ``mind_ctx`` bound to ``device`` and the current torch stream; ``load_state_dict`` mirrors
    ``ScenePredNet.load_state_dict`` (reference planners/mind/planner.py:46-48), ``predict`` mirrors
    ``ScenePredNet.forward`` over a collated batch (planners/mind/networks/network.py:582-595)."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("HipPredictor needs a GPU (no CPU fallback)")
        self.device = torch.device("cuda", device if isinstance(device, int) else (device.index or 0))
        torch.cuda.set_device(self.device)
        self.ctx = C.c_void_p()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        # the legacy default stream couples every synchronisation and kernel of this context to the other streams of the process
        # (measured: the device-side window assembly of mind_aime_rebase then stalls for 10-25 ms on entry)
        self.on_default_stream = int(stream or 0) == 0
        rc = self.lib.mind_ctx_create(self.device.index, C.c_void_p(stream), C.byref(self.ctx))
        _lib.check(self.lib, None, rc, "mind_ctx_create")
        self._keep = []

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.mind_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd):
        descs = (_lib.TensorDesc * len(sd))()
        keep = []
        for i, (k, v) in enumerate(sd.items()):
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            keep.append(a)
            descs[i].name = k.encode()
            descs[i].data = a.ctypes.data_as(C.POINTER(C.c_float))
            descs[i].numel = a.size
        rc = self.lib.mind_weights_load(self.ctx, descs, len(sd))
        _lib.check(self.lib, self.ctx, rc, "mind_weights_load")

    PAIR_PREC = ("f32", "bf16x3", "bf16", "bf16x6")

    def pair_precision(self):
        """arithmetic of the predictor's MFMA contractions: 'bf16x6' (default: exact three-way split, fp32 class) | 'f32' | 'bf16x3' | 'bf16' (include/mind_hip.h)"""
        return self.PAIR_PREC[self.lib.mind_get_pair_precision(self.ctx)]

    def set_tuning(self, name, value):
        """kernel-selection knobs of include/mind_hip.h (mind_set_tuning): dec_mfma_min, enc_mfma, actor_split, xcd_order"""
        _lib.check(self.lib, self.ctx, self.lib.mind_set_tuning(self.ctx, name.encode(), int(value)), "mind_set_tuning")
        if name == "ilqr_wgs":
            self._ilqr_wgs_user = self._ilqr_wgs_now = int(value)

    def set_pair_precision(self, name):
        _lib.check(self.lib, self.ctx, self.lib.mind_set_pair_precision(self.ctx, self.PAIR_PREC.index(name)), "mind_set_pair_precision")

    def set_profiling(self, on):
        self.lib.mind_set_profiling(self.ctx, 1 if on else 0)

    def ilqr_stats(self):
        """(kernel ms, cost trees, workgroups per tree) of the last tree-iLQR call on this context (ms = 0 unless profiling)"""
        ms, n, g = C.c_float(), C.c_int(), C.c_int()
        self.lib.mind_last_ilqr_stats(self.ctx, C.byref(ms), C.byref(n), C.byref(g))
        return ms.value, n.value, g.value

    def ilqr_profile(self):
        """phase cycles of the critical cost tree of the last tree-iLQR launch (mind_last_ilqr_profile)"""
        v = (C.c_double * 9)()
        self.lib.mind_last_ilqr_profile(self.ctx, v)
        keys = ("nodes", "depth", "passes", "derivatives", "backward", "state_chain", "cost_pass", "selection", "trees")
        return dict(zip(keys, [float(x) for x in v]))

    def ilqr_trace(self, tree, phase=0):
        """[iterations, 4] float64 per-iteration trace of one fit of the last tree-iLQR call (mind_last_ilqr_trace): mu, J of the nominal
        trajectory, accepted step index (-1 rejected, -2 singular Q_uu), J of the accepted candidate"""
        n = C.c_int()
        buf = np.zeros((256, 4), np.float64)
        _lib.check(self.lib, self.ctx, self.lib.mind_last_ilqr_trace(self.ctx, int(tree), int(phase), buf.ctypes.data_as(C.POINTER(C.c_double)), 256,
                                                                    C.byref(n)), "mind_last_ilqr_trace")
        return buf[:min(n.value, 256)].copy()

    def fusion_stats(self):
        n = C.c_int()
        ms = C.c_float()
        pairs = C.c_double()
        self.lib.mind_last_fusion_stats(self.ctx, C.byref(n), C.byref(ms), C.byref(pairs))
        return n.value, ms.value, pairs.value

    def debug_set_layers(self, n):
        _lib.check(self.lib, self.ctx, self.lib.mind_debug_set_layers(self.ctx, n), "mind_debug_set_layers")

    def debug_read(self, name):
        n = self.lib.mind_debug_read(self.ctx, name.encode(), None, 0)
        if n < 0:
            raise _lib.MindError(f"mind_debug_read({name}) -> {n}")
        buf = np.empty(n, np.float32)
        got = self.lib.mind_debug_read(self.ctx, name.encode(), buf.ctypes.data_as(C.POINTER(C.c_float)), n)
        assert got == n, (got, n)
        return buf

    def synchronize(self):
        _lib.check(self.lib, self.ctx, self.lib.mind_ctx_synchronize(self.ctx), "mind_ctx_synchronize")

    def predict(self, actors, actor_off, lanes, lane_off, actor_ctrs, actor_vecs, lane_ctrs, lane_vecs,
                tgt_nodes, tgt_rpe, rpe=None, lane_feat=None, want_lane_feat=False, taps=False):
        """All tensors fp32 contiguous on ``self.device``; actor_off/lane_off are int prefix-sum lists [B+1].
        Returns dict(cls [B,6], reg [A,6,60,5], vel [A,6,60,2], ...)."""
        dev = self.device
        B = len(actor_off) - 1
        A, L = int(actor_off[-1]), int(lane_off[-1])

        def f32(t):
            if t is None:
                return None
            assert t.device == dev and t.dtype == torch.float32, (t.device, t.dtype)
            return t.contiguous()

        actors, lanes, lane_feat = f32(actors), f32(lanes), f32(lane_feat)
        actor_ctrs, actor_vecs, lane_ctrs, lane_vecs = f32(actor_ctrs), f32(actor_vecs), f32(lane_ctrs), f32(lane_vecs)
        tgt_nodes, tgt_rpe = f32(tgt_nodes), f32(tgt_rpe)
        assert actors.shape == (A, 14, 48) and tgt_nodes.shape == (B, 10, 16) and tgt_rpe.shape == (B, 20)
        ao = (C.c_int32 * (B + 1))(*[int(v) for v in actor_off])
        lo = (C.c_int32 * (B + 1))(*[int(v) for v in lane_off])
        sb = _lib.SceneBatch()
        sb.n_scenes = B
        sb.actor_off, sb.lane_off = ao, lo

        def ptr(t):
            return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else None

        sb.actors, sb.lanes, sb.lane_feat = ptr(actors), ptr(lanes), ptr(lane_feat)
        sb.actor_ctrs, sb.actor_vecs, sb.lane_ctrs, sb.lane_vecs = ptr(actor_ctrs), ptr(actor_vecs), ptr(lane_ctrs), ptr(lane_vecs)
        rpe_keep = None
        if rpe is not None:
            rpe_keep = [f32(r) for r in rpe]
            arr = (C.c_void_p * B)(*[r.data_ptr() for r in rpe_keep])
            sb.rpe = arr
        sb.tgt_nodes, sb.tgt_rpe = ptr(tgt_nodes), ptr(tgt_rpe)
        out = {"cls": torch.empty(B, 6, device=dev), "reg": torch.empty(A, 6, 60, 5, device=dev),
               "vel": torch.empty(A, 6, 60, 2, device=dev)}
        po = _lib.PredOut()
        po.cls, po.reg, po.vel = ptr(out["cls"]), ptr(out["reg"]), ptr(out["vel"])
        if want_lane_feat and lane_feat is None and L > 0:
            out["lane_feat"] = torch.empty(L, 128, device=dev)
            po.lane_feat = ptr(out["lane_feat"])
        if taps:
            out["actor_emb"] = torch.empty(A, 128, device=dev)
            out["cls_emb"] = torch.empty(B, 128, device=dev)
            po.actor_emb, po.cls_emb = ptr(out["actor_emb"]), ptr(out["cls_emb"])
        rc = self.lib.mind_predict_batch(self.ctx, C.byref(sb), C.byref(po))
        _lib.check(self.lib, self.ctx, rc, "mind_predict_batch")
        return out

    # ------------------------------------------------------------------------------------------
    def ilqr_contingency(self, cfg_warm, cfg_full, flats, x0, lane, target_vel):
        """Warm-start fit + full fit of all cost trees of a plan in one launch (mind_ilqr_contingency).
        Returns (xs list[[M,6]], us list[[M,2]], stats_warm list[dict], stats_full list[dict])."""
        call = IlqrCall(self.lib, cfg_warm, flats, x0, lane, target_vel, cfg_full=cfg_full)
        call.run(self)
        return call.finish()

    def ilqr_solve(self, cfg, flats, x0, lane, target_vel, use_exo, us_init=None):
        """Solve all cost trees of a plan in one launch.  ``flats``: list of dicts with parent int32 [M],
        prob f32 [M], mean f32 [M,a,2], cov f32 [M,a] (trajectory-node arrays in creation order);
        ``cfg``: ``_lib.IlqrCfg``.  Returns (xs list[[M,6]], us list[[M,2]], stats list[dict])."""
        call = IlqrCall(self.lib, cfg, flats, x0, lane, target_vel, use_exo=use_exo, us_init=us_init)
        call.run(self)
        return call.finish()

    # ---- planners/ilqr surface: arbitrary materialised fields + per-node quadratic potentials -------------
    @staticmethod
    def _generic_tree(tree, keep):
        """tree: dict(parent int32 [M], field f64 [M,H,W], node_w f64 [M,32]) -> CostTree."""
        ct = _lib.CostTree()
        par = np.ascontiguousarray(tree["parent"], np.int32)
        fld = np.ascontiguousarray(tree["field"], np.float64)
        nw = np.ascontiguousarray(tree["node_w"], np.float64)
        if fld.ndim != 3 or fld.shape[0] != len(par) or nw.shape != (len(par), 32):
            raise ValueError("generic cost tree: field [M,H,W] / node_w [M,32] do not match parent [M]")
        keep += [par, fld, nw]
        ct.n_nodes, ct.n_agents = len(par), 0
        ct.parent = par.ctypes.data_as(C.POINTER(C.c_int32))
        ct.field = fld.ctypes.data_as(C.POINTER(C.c_double))
        ct.node_w = nw.ctypes.data_as(C.POINTER(C.c_double))
        return ct

    @staticmethod
    def _grid(grid, keep):
        """grid: dict(offset [2], res, gx [W], gy [H]) as PotentialField receives them."""
        g = _lib.FieldGrid()
        gx = np.ascontiguousarray(grid["gx"], np.float64)
        gy = np.ascontiguousarray(grid["gy"], np.float64)
        keep += [gx, gy]
        g.W, g.H, g.res = len(gx), len(gy), float(grid["res"])
        g.off_x, g.off_y = float(grid["offset"][0]), float(grid["offset"][1])
        g.gx = gx.ctypes.data_as(C.POINTER(C.c_double))
        g.gy = gy.ctypes.data_as(C.POINTER(C.c_double))
        return g

    def ilqr_solve_fields(self, cfg, grid, tree, x0, us_init=None):
        """iLQR.fit on ONE tree with materialised fields (mind_ilqr_solve_fields).  Returns (xs [M,6], us [M,2], stats)."""
        keep = []
        ct = self._generic_tree(tree, keep)
        g = self._grid(grid, keep)
        M = ct.n_nodes
        x0 = np.ascontiguousarray(x0, np.float64)
        xs, us = np.zeros((M, 6)), np.zeros((M, 2))
        st = _lib.IlqrStats()
        ui = None if us_init is None else np.ascontiguousarray(us_init, np.float64)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
        rc = self.lib.mind_ilqr_solve_fields(self.ctx, C.byref(cfg), C.byref(g), C.byref(ct), 1, dp(x0), dp(ui), dp(xs), dp(us),
                                             C.byref(st))
        _lib.check(self.lib, self.ctx, rc, "mind_ilqr_solve_fields")
        return xs, us, dict(iterations=st.iterations, converged=st.converged, J=st.J, mu=st.mu)

    def cost_eval(self, cfg, node, x, u, tree, grid=None, x0=None, lane=None, target_vel=0.0, use_exo=0):
        """TreeCost.l/l_x/l_u/l_xx/l_uu of ``tree`` at (x[q], u[q]) for node[q] (mind_cost_eval).
        grid given: generic tree dict; grid None: planner-mode flat dict (parent, prob, mean, cov) + x0, lane.
        Returns dict(l [Q], l_x [Q,6], l_u [Q,2], l_xx [Q,6,6], l_uu [Q,2,2])."""
        keep = []
        if grid is not None:
            ct = self._generic_tree(tree, keep)
            g = C.byref(self._grid(grid, keep))
            x0 = np.zeros(6) if x0 is None else x0
        else:
            ct = _lib.CostTree()
            par = np.ascontiguousarray(tree["parent"], np.int32)
            prob = np.ascontiguousarray(tree["prob"], np.float32)
            mean = np.ascontiguousarray(tree["mean"], np.float32)
            cov = np.ascontiguousarray(tree["cov"], np.float32)
            keep += [par, prob, mean, cov]
            ct.n_nodes, ct.n_agents = len(par), mean.shape[1]
            ct.parent = par.ctypes.data_as(C.POINTER(C.c_int32))
            ct.prob = prob.ctypes.data_as(C.POINTER(C.c_float))
            ct.agent_mean = mean.ctypes.data_as(C.POINTER(C.c_float))
            ct.agent_cov = cov.ctypes.data_as(C.POINTER(C.c_float))
            g = None
        node = np.ascontiguousarray(node, np.int32)
        x = np.ascontiguousarray(x, np.float64).reshape(len(node), 6)
        u = np.ascontiguousarray(u, np.float64).reshape(len(node), 2)
        x0 = np.ascontiguousarray(x0, np.float64)
        lane_a = None if lane is None else np.ascontiguousarray(lane, np.float64)
        out = np.zeros((len(node), 47))
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
        rc = self.lib.mind_cost_eval(self.ctx, C.byref(cfg), g, C.byref(ct), dp(x0), dp(lane_a), 0 if lane_a is None else len(lane_a),
                                     C.c_double(float(target_vel)), int(use_exo), len(node),
                                     node.ctypes.data_as(C.POINTER(C.c_int32)), dp(x), dp(u), dp(out))
        _lib.check(self.lib, self.ctx, rc, "mind_cost_eval")
        luu = np.zeros((len(node), 2, 2))
        luu[:, 0, 0], luu[:, 1, 1] = out[:, 45], out[:, 46]
        return dict(l=out[:, 0].copy(), l_x=out[:, 1:7].copy(), l_u=out[:, 7:9].copy(), l_xx=out[:, 9:45].reshape(-1, 6, 6).copy(), l_uu=luu)

    def aime_world(self, reg, vel, actor_ctrs, actor_vecs, a_off, rots, origs, cov_last, last, target_lane=None, cls=None,
                   scen_prob=None, dist_thres=None):
        """k7 on the device (mind_aime_world): reg [A,6,60,5] / vel [A,6,60,2] / actor_ctrs, actor_vecs [A,2] device
        tensors; a_off [B+1]; rots [B,2,2], origs [B,2], cov_last [A] host float32; last [B] int.
        target_lane [P,2] float32 (optional).  Returns device tensors world [A,6,60,6] (x,y,vx,vy,heading,max-sigma),
        topo [A,6], ego_end [B,6,4] (ego x, y, max-sigma at step `last`, distance of that point to the target lane).
        With ``cls`` [B,6] (device) and ``scen_prob`` [B] the pruning decisions are taken on the device as well (k_aime_select):
        ``sel`` [2,B,6] = (mode index or -1, path probability) of the kept modes in visiting order; ``dist_thres`` switches the
        target-lane test on (needs target_lane and last >= 0 everywhere)."""
        dev = self.device
        B, A = len(a_off) - 1, int(a_off[-1])
        for t_ in (reg, vel, actor_ctrs, actor_vecs):
            assert t_.device == dev and t_.dtype == torch.float32 and t_.is_contiguous()
        assert reg.shape == (A, 6, 60, 5) and vel.shape == (A, 6, 60, 2) and actor_ctrs.shape == (A, 2)
        wi, wo = _lib.WorldIn(), _lib.WorldOut()
        ao = (C.c_int32 * (B + 1))(*[int(v) for v in a_off])
        rot = np.ascontiguousarray(rots, np.float32).reshape(B, 4)
        org = np.ascontiguousarray(origs, np.float32).reshape(B, 2)
        cvl = np.ascontiguousarray(cov_last, np.float32).reshape(A)
        lst = np.ascontiguousarray(last, np.int32).reshape(B)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        wi.n_scenes, wi.actor_off = B, ao
        wi.reg, wi.vel = C.c_void_p(reg.data_ptr()), C.c_void_p(vel.data_ptr())
        wi.actor_ctrs, wi.actor_vecs = C.c_void_p(actor_ctrs.data_ptr()), C.c_void_p(actor_vecs.data_ptr())
        wi.rot, wi.orig, wi.cov_last = fp(rot), fp(org), fp(cvl)
        wi.last = lst.ctypes.data_as(C.POINTER(C.c_int32))
        if target_lane is not None:
            tl = np.ascontiguousarray(target_lane, np.float32).reshape(-1, 2)
            wi.target_lane, wi.n_lane_pts = fp(tl), len(tl)
        # topo and ego_end live in one buffer behind a copy of cls, so that the host fetches all three with one copy
        if cls is not None and scen_prob is not None:
            small = torch.empty(B * 6 + A * 6 + B * 24, device=dev)     # the host only reads `sel`
        else:
            small = torch.zeros(B * 6 + A * 6 + B * 24, device=dev)
            if cls is not None:
                small[:B * 6] = cls.reshape(-1)
        out = dict(world=torch.empty(A, 6, 60, 6, device=dev), small=small, topo=small[B * 6:B * 6 + A * 6].view(A, 6),
                   ego_end=small[B * 6 + A * 6:].view(B, 6, 4))
        wo.world, wo.topo, wo.ego_end = (C.c_void_p(out[k].data_ptr()) for k in ("world", "topo", "ego_end"))
        if cls is not None and scen_prob is not None:
            cls = cls.contiguous()
            assert cls.device == dev and cls.dtype == torch.float32 and cls.numel() == B * 6
            out["_cls"] = cls                       # keeps the (possibly fresh) buffer alive until the kernel ran
            sp = np.ascontiguousarray(scen_prob, np.float32).reshape(B)
            out["sel"] = torch.empty(2, B, 6, device=dev)
            wi.cls, wi.scen_prob = C.c_void_p(cls.data_ptr()), fp(sp)
            wi.lane_check = int(dist_thres is not None)
            wi.dist_thres = float(dist_thres) if dist_thres is not None else 0.0
            wo.sel, wo.sel_prob = C.c_void_p(out["sel"][0].data_ptr()), C.c_void_p(out["sel"][1].data_ptr())
        rc = self.lib.mind_aime_world(self.ctx, C.byref(wi), C.byref(wo))
        _lib.check(self.lib, self.ctx, rc, "mind_aime_world")
        return out

    def aime_rebase(self, pos, ang, vel, types, lane_ctrs, lane_vecs, target_lane, target_lane_info, pad=None,
                    time_ahead=5.0, min_vel=0.5, dev_src=None):
        """update_obser for S child scenes on the device (mind_aime_rebase).  pos [S,a,50,2], ang [S,a,50], vel [S,a,50,2]
        (world-frame windows), types [a,50,7], lane_ctrs / lane_vecs [l,2], target_lane [P,2], target_lane_info [P,12]: host
        float32.  Returns device tensors actors [S*a,14,48], actor_ctrs, actor_vecs [S*a,2], lane_ctrs, lane_vecs [S*l,2],
        tgt_nodes [S,10,16], tgt_rpe [S,20], frames [S,28] (ROT, ORIG, TGT_PTS) and ``gen``, the call's generation.
        ``dev_src`` = dict(rows=device [R,60,6], parent_slot, row0, dur: int [S], gen, a): the windows are assembled on the device
        from the parents' windows of the previous call (generation ``gen``) and the kept rows; pos / ang / vel are ignored."""
        dev = self.device
        f = lambda x: np.ascontiguousarray(x, np.float32)
        types = f(types)
        lane_ctrs, lane_vecs, tl, ti = f(lane_ctrs), f(lane_vecs), f(target_lane), f(target_lane_info)
        ri, ro = _lib.RebaseIn(), _lib.RebaseOut()
        fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
        ip = lambda x: x.ctypes.data_as(C.POINTER(C.c_int32))
        keep = []
        if dev_src is not None:
            rows = dev_src["rows"]
            assert rows.device == dev and rows.dtype == torch.float32 and rows.is_contiguous() and tuple(rows.shape[1:]) == (60, 6)
            ps, r0, du = (np.ascontiguousarray(dev_src[k], np.int32) for k in ("parent_slot", "row0", "dur"))
            S, a = len(ps), int(dev_src["a"])
            assert len(r0) == S and len(du) == S and int(r0.max()) + a <= rows.shape[0]
            ri.rows_dev, ri.parent_slot, ri.row0, ri.dur, ri.prev_gen = C.c_void_p(rows.data_ptr()), ip(ps), ip(r0), ip(du), int(dev_src["gen"])
            keep += [ps, r0, du, rows]
        else:
            # lists of per-scene windows are stacked here (page-locked staging buffers were measured slower than the runtime's own
            # pageable path: 15 vs 10 ms per plan at 216 scenes x 64 agents)
            stk = lambda x: f(np.stack(x) if isinstance(x, (list, tuple)) else x)
            pos, ang, vel = stk(pos), stk(ang), stk(vel)
            S, a = pos.shape[:2]
            assert pos.shape == (S, a, 50, 2) and ang.shape == (S, a, 50) and vel.shape == (S, a, 50, 2)
            ri.pos, ri.ang, ri.vel = fp(pos), fp(ang), fp(vel)
        l = lane_ctrs.shape[0]
        assert types.shape == (a, 50, 7) and tl.ndim == 2 and ti.shape == (len(tl), 12)
        ri.n_scenes, ri.n_agents, ri.n_lanes = S, a, l
        ri.types = fp(types)
        padk = None
        if pad is not None:
            padk = f(pad)
            ri.pad = fp(padk)
        ri.lane_ctrs, ri.lane_vecs, ri.target_lane, ri.target_lane_info, ri.n_lane_pts = fp(lane_ctrs), fp(lane_vecs), fp(tl), fp(ti), len(tl)
        ri.time_ahead, ri.min_vel = float(time_ahead), float(min_vel)
        out = dict(actors=torch.empty(S * a, 14, 48, device=dev), actor_ctrs=torch.empty(S * a, 2, device=dev),
                   actor_vecs=torch.empty(S * a, 2, device=dev), lane_ctrs=torch.empty(S * l, 2, device=dev),
                   lane_vecs=torch.empty(S * l, 2, device=dev), tgt_nodes=torch.empty(S, 10, 16, device=dev),
                   tgt_rpe=torch.empty(S, 20, device=dev), frames=torch.empty(S, 28, device=dev))
        for k in out:
            setattr(ro, k, C.c_void_p(out[k].data_ptr()))
        gen = C.c_int32(0)
        ro.gen = C.pointer(gen)
        rc = self.lib.mind_aime_rebase(self.ctx, C.byref(ri), C.byref(ro))
        _lib.check(self.lib, self.ctx, rc, "mind_aime_rebase")
        out["gen"] = self._rebase_gen = int(gen.value)      # the arena of THIS call is what a device-source call can build on next
        return out

    def aime_plan(self, root, hist, lane_ctrs, lane_vecs, target_lane, target_lane_info, time_ahead, dist_thres, max_depth,
                  pred_len=50, min_vel=0.5, max_rounds=16, raw=None, script=None, prob_floor=None, solve=None):
        """ScenarioTreeGenerator.branch_aime in one call (mind_aime_plan).  Host-built root: ``root`` = the root scene dict of
        process_data (ACTORS, TRAJS_CTRS, TRAJS_VECS, LANES, TGT_NODES, TGT_RPE, ROT, ORIG, TGT_PTS, TRAJS_TYPE), ``hist`` [a,50,6] its
        world-frame history (x, y, vx, vy, heading, max-sigma), lane_ctrs / lane_vecs the lane graph's anchors.  Device-built root:
        ``raw`` = dict(pos [a,50,2], ang [a,50], vel [a,50,2], pad [a,50], types [a,50,7] as get_agent_trajectories returns them,
        lane_pts [l,11,2] float64, lane_flags [l,6] int, travel0) and root / hist / lane_ctrs / lane_vecs are ignored.
        Returns (nodes: structured array, one record per internal tree node in creation order, rows: float32 [n], info dict) or None
        when the library reports a situation only the round-by-round path handles.  ``script``: see mind_aime_plan_in.script_cls."""
        pi, keep, a, l = self._aime_plan_args(root, hist, lane_ctrs, lane_vecs, target_lane, target_lane_info, time_ahead, dist_thres, max_depth,
                                              pred_len, min_vel, max_rounds, raw, script, prob_floor, solve)
        po = _lib.AimePlanOut()
        rc = self.lib.mind_aime_plan(self.ctx, C.byref(pi), C.byref(po))
        return self._aime_plan_result(rc, po, a, l)

    def aime_plan_begin(self, *args, **kw):
        """aime_plan in two halves (mind_aime_plan_begin / _finish: the plan runs on a thread of the library, this call returns at once):
        same arguments as aime_plan; aime_plan_ready() tells whether aime_plan_finish() would block."""
        pi, keep, a, l = self._aime_plan_args(*args, **kw)
        rc = self.lib.mind_aime_plan_begin(self.ctx, C.byref(pi))
        _lib.check(self.lib, self.ctx, rc, "mind_aime_plan_begin")
        self._aime_pending = (pi, keep, a, l)         # (the input arrays stay alive until the plan is collected)

    def aime_plan_ready(self):
        return self.lib.mind_aime_plan_poll(self.ctx) == 0

    def aime_plan_finish(self):
        pi, keep, a, l = self._aime_pending
        self._aime_pending = None
        po = _lib.AimePlanOut()
        rc = self.lib.mind_aime_plan_finish(self.ctx, C.byref(po))
        return self._aime_plan_result(rc, po, a, l)

    def busy(self):
        """work queued on this context (a begun plan, begun contingency solves) has not completed yet"""
        rc = self.lib.mind_ctx_busy(self.ctx)
        if rc < 0:
            _lib.check(self.lib, self.ctx, rc, "mind_ctx_busy")
        return rc == 1

    def _aime_plan_args(self, root, hist, lane_ctrs, lane_vecs, target_lane, target_lane_info, time_ahead, dist_thres, max_depth,
                        pred_len=50, min_vel=0.5, max_rounds=16, raw=None, script=None, prob_floor=None, solve=None):
        f = lambda x: np.ascontiguousarray(x, np.float32)
        fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
        pi = _lib.AimePlanIn()
        keep = []
        if raw is not None:
            arrs = dict(raw_pos=f(raw["pos"]), raw_ang=f(raw["ang"]), raw_vel=f(raw["vel"]), raw_pad=f(raw["pad"]), types=f(raw["types"]),
                        target_lane=f(target_lane), target_lane_info=f(target_lane_info))
            lpts = np.ascontiguousarray(raw["lane_pts"], np.float64)
            lfl = np.ascontiguousarray(raw["lane_flags"], np.int32)
            a, l, P = arrs["raw_pos"].shape[0], lpts.shape[0], arrs["target_lane"].shape[0]
            assert arrs["raw_pos"].shape == (a, 50, 2) and arrs["raw_ang"].shape == (a, 50) and arrs["raw_pad"].shape == (a, 50)
            assert arrs["types"].shape == (a, 50, 7) and lpts.shape == (l, 11, 2) and lfl.shape == (l, 6)
            pi.lane_pts, pi.lane_flags = lpts.ctypes.data_as(C.POINTER(C.c_double)), lfl.ctypes.data_as(C.POINTER(C.c_int32))
            pi.travel0 = float(raw["travel0"])
            keep += [lpts, lfl]
        else:
            arrs = dict(actors=f(root["ACTORS"]), actor_ctrs=f(root["TRAJS_CTRS"]), actor_vecs=f(root["TRAJS_VECS"]), lanes=f(root["LANES"]),
                        lane_ctrs=f(lane_ctrs), lane_vecs=f(lane_vecs), tgt_nodes=f(root["TGT_NODES"]), tgt_rpe=f(root["TGT_RPE"]),
                        rot=f(root["ROT"]), orig=f(root["ORIG"]), tgt_pts=f(root["TGT_PTS"]), hist=f(hist), types=f(root["TRAJS_TYPE"]),
                        target_lane=f(target_lane), target_lane_info=f(target_lane_info))
            a, l, P = arrs["actors"].shape[0], arrs["lanes"].shape[0], arrs["target_lane"].shape[0]
            assert arrs["actors"].shape == (a, 14, 48) and arrs["hist"].shape == (a, 50, 6) and arrs["types"].shape == (a, 50, 7)
            assert arrs["lanes"].shape == (l, 10, 16) and arrs["lane_ctrs"].shape == (l, 2)
        assert arrs["target_lane_info"].shape == (P, 12)
        pi.n_agents, pi.n_lanes, pi.n_lane_pts = a, l, P
        for k, v in arrs.items():
            setattr(pi, k, fp(v))
        pi.time_ahead, pi.min_vel, pi.dist_thres = float(time_ahead), float(min_vel), float(dist_thres)
        pi.max_depth, pi.max_rounds, pi.pred_len = int(max_depth), int(max_rounds), int(pred_len)
        pi.prob_floor = 0.0 if prob_floor is None else float(prob_floor)      # (0 = the reference's 0.001)
        if script is not None:        # (cls [1,6], reg [a,6,60,5], vel [a,6,60,2]) float32 device tensors: scripted modes (synth.ScriptedBranching)
            sc, sr, sv = script
            assert tuple(sr.shape) == (a, 6, 60, 5) and tuple(sv.shape) == (a, 6, 60, 2) and sc.numel() == 6
            assert all(t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 for t in script)
            pi.script_cls, pi.script_reg, pi.script_vel = sc.data_ptr(), sr.data_ptr(), sv.data_ptr()
            keep += list(script)
        if solve is not None:         # (cfg_warm, cfg_full, x0 [6], lane [P,2], target velocity): the plan begins its contingency solves itself
            cw, cf, x0, lane_s, tv = solve
            x0 = np.ascontiguousarray(x0, np.float64)
            lane_s = np.ascontiguousarray(lane_s, np.float64)
            pi.solve_cfg_warm, pi.solve_cfg_full = C.addressof(cw), C.addressof(cf)
            pi.solve_x0, pi.solve_lane, pi.solve_n_lane_pts, pi.solve_target_vel = x0.ctypes.data, lane_s.ctypes.data, len(lane_s), float(tv)
            keep += [cw, cf, x0, lane_s]
        keep.append(arrs)
        return pi, keep, a, l

    def _aime_plan_result(self, rc, po, a, l):
        if rc == _lib.MIND_ESTATE:
            msg = self.lib.mind_last_error_string(self.ctx) or b""
            if msg.startswith(b"unsupported"):
                self.last_aime_fallback = msg.decode()
                return None
        _lib.check(self.lib, self.ctx, rc, "mind_aime_plan")
        n = po.n_nodes
        nodes = np.frombuffer(C.string_at(po.nodes, n * C.sizeof(_lib.AimeNode)), dtype=AIME_NODE_DTYPE) if n else np.zeros(0, AIME_NODE_DTYPE)
        nf = int(po.n_row_floats)
        rows = np.frombuffer(C.string_at(po.rows, nf * 4), dtype=np.float32) if nf else np.zeros(0, np.float32)
        # the flattened cost trees (what flatten_scenario_tree builds from the returned scenario trees), one dict per scenario tree
        flats, nt = [], po.n_trees
        if nt:
            off = np.frombuffer(C.string_at(po.tree_off, (nt + 1) * 4), dtype=np.int32)
            top = np.frombuffer(C.string_at(po.tree_top, nt * 4), dtype=np.int32)
            Mt = int(off[-1])
            par = np.frombuffer(C.string_at(po.flat_parent, Mt * 4), dtype=np.int32)
            prob = np.frombuffer(C.string_at(po.flat_prob, Mt * 4), dtype=np.float32)
            mean = np.frombuffer(C.string_at(po.flat_mean, Mt * a * 8), dtype=np.float32).reshape(Mt, a, 2)
            cov = np.frombuffer(C.string_at(po.flat_cov, Mt * a * 4), dtype=np.float32).reshape(Mt, a)
            for t in range(nt):
                lo, hi = int(off[t]), int(off[t + 1])
                flats.append((int(top[t]), dict(parent=par[lo:hi], prob=prob[lo:hi], mean=mean[lo:hi], cov=cov[lo:hi])))
        info = dict(n_expanded=po.n_expanded, n_rounds=po.n_rounds, root_flags=po.root_flags, a=a, l=l, flats=flats,
                    round_scenes=[po.round_scenes[i] for i in range(po.n_rounds)], pair_ms=po.pair_ms, pair_launches=po.pair_launches,
                    solves_begun=bool(po.solves_begun))
        self.last_aime_info = info
        return nodes, rows, info

    def lane_dist_field(self, ego_xy, lane, W, H, res):
        """gen_dist_field (ilqr/utils.py:5-22) -> (offset [2], gx [W], gy [H], dist [H,W])."""
        ego = np.ascontiguousarray(np.asarray(ego_xy, np.float64)[:2])
        lane = np.ascontiguousarray(lane, np.float64)
        off, gx, gy, dist = np.zeros(2), np.zeros(W), np.zeros(H), np.zeros((H, W))
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        rc = self.lib.mind_lane_dist_field(self.ctx, dp(ego), dp(lane), len(lane), int(W), int(H), C.c_double(float(res)),
                                           dp(off), dp(gx), dp(gy), dp(dist))
        _lib.check(self.lib, self.ctx, rc, "mind_lane_dist_field")
        return off, gx, gy, dist

    def predict_numpy_batch(self, pb, use_rpe=False, **kw):
        """Convenience for tests: ``pb`` as produced by ``mind_amd.synth.predictor_batch`` (numpy)."""
        dev = self.device
        B = len(pb["ACTOR_IDCS"])
        a_off = [0] + list(np.cumsum([len(x) for x in pb["ACTOR_IDCS"]]))
        l_off = [0] + list(np.cumsum([len(x) for x in pb["LANE_IDCS"]]))
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
        actr = np.concatenate([pb["CTRS"][b][:a_off[b + 1] - a_off[b]] for b in range(B)])
        avec = np.concatenate([pb["VECS"][b][:a_off[b + 1] - a_off[b]] for b in range(B)])
        lctr = np.concatenate([pb["CTRS"][b][a_off[b + 1] - a_off[b]:] for b in range(B)])
        lvec = np.concatenate([pb["VECS"][b][a_off[b + 1] - a_off[b]:] for b in range(B)])
        rpe = None
        if use_rpe:
            rpe = [t(r) for r in pb["RPE"]]
        return self.predict(t(pb["ACTORS"]), a_off, t(pb["LANES"]), l_off, t(actr), t(avec), t(lctr), t(lvec),
                            t(pb["TGT_NODES"]), t(pb["TGT_RPE"]), rpe=rpe, **kw)
