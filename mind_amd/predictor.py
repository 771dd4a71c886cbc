"""Host-side handle on the HIP scene predictor (one context per process/GPU)."""
import ctypes as C

import numpy as np
import torch

from . import _lib


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

    def set_profiling(self, on):
        self.lib.mind_set_profiling(self.ctx, 1 if on else 0)

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
    def ilqr_solve(self, cfg, flats, x0, lane, target_vel, use_exo, us_init=None):
        """Solve all cost trees of a plan in one launch.  ``flats``: list of dicts with parent int32 [M],
        prob f32 [M], mean f32 [M,a,2], cov f32 [M,a] (trajectory-node arrays in creation order);
        ``cfg``: ``_lib.IlqrCfg``.  Returns (xs list[[M,6]], us list[[M,2]], stats list[dict])."""
        n = len(flats)
        trees = (_lib.CostTree * n)()
        keep = []
        Ms = []
        for i, f in enumerate(flats):
            par = np.ascontiguousarray(f["parent"], np.int32)
            prob = np.ascontiguousarray(f["prob"], np.float32)
            mean = np.ascontiguousarray(f["mean"], np.float32)
            cov = np.ascontiguousarray(f["cov"], np.float32)
            keep += [par, prob, mean, cov]
            trees[i].n_nodes = len(par)
            trees[i].parent = par.ctypes.data_as(C.POINTER(C.c_int32))
            trees[i].prob = prob.ctypes.data_as(C.POINTER(C.c_float))
            trees[i].n_agents = mean.shape[1]
            trees[i].agent_mean = mean.ctypes.data_as(C.POINTER(C.c_float))
            trees[i].agent_cov = cov.ctypes.data_as(C.POINTER(C.c_float))
            Ms.append(len(par))
        Mt = int(sum(Ms))
        x0 = np.ascontiguousarray(x0, np.float64)
        lane = np.ascontiguousarray(lane, np.float64)
        xs = np.zeros((Mt, 6))
        us = np.zeros((Mt, 2))
        st = (_lib.IlqrStats * n)()
        ui = None if us_init is None else np.ascontiguousarray(np.concatenate(us_init), np.float64)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
        rc = self.lib.mind_ilqr_solve_trees(self.ctx, C.byref(cfg), trees, n, dp(x0), dp(lane), len(lane),
                                            C.c_double(float(target_vel)), int(use_exo), dp(ui), dp(xs), dp(us), st)
        _lib.check(self.lib, self.ctx, rc, "mind_ilqr_solve_trees")
        offs = np.cumsum([0] + Ms)
        return ([xs[offs[i]:offs[i + 1]] for i in range(n)], [us[offs[i]:offs[i + 1]] for i in range(n)],
                [dict(iterations=st[i].iterations, converged=st[i].converged, J=st[i].J, mu=st[i].mu) for i in range(n)])

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
