"""``ScenePredNet``: the reference's predictor call surface (planners/mind/networks/network.py:559-606)
over the hand-written HIP kernels.  ``pre_process(data)`` / ``__call__(data_in)`` / ``load_state_dict`` /
``to`` / ``eval`` behave like the torch module they replace; there is no torch compute inside."""
import numpy as np
import torch

from ....runtime import get_runtime


class ScenePredNet:
    computes_rpe_in_kernel = True   # RPE (mind/utils.py:193-212) is evaluated inside the fusion kernel

    def __init__(self, cfg=None, device=None, own_context=False):
        self.cfg = cfg or {}
        self.device = device
        idx = device.index if isinstance(device, torch.device) and device.index is not None else 0
        # own_context: a context / stream of its own instead of the thread's (several planners driven from one thread, runtime.new_runtime)
        if own_context:
            from ....runtime import new_runtime
            self.rt = new_runtime(idx)
        else:
            self.rt = get_runtime(idx)
        self.last_lane_feat = None
        self.last_packed = None
        self._loaded = False

    # torch.nn.Module look-alikes -----------------------------------------------------------------
    def load_state_dict(self, sd, strict=True):
        self.rt.load_state_dict(sd)
        self._loaded = True
        return self

    def to(self, device):
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(())

    # ---------------------------------------------------------------------------------------------
    def pre_process(self, data):
        """Host batch dict (collate layout, mind/utils.py:142-168) -> device tensors + offsets."""
        dev = self.rt.device
        a_off = [0]
        for i in data["ACTOR_IDCS"]:
            a_off.append(a_off[-1] + len(i))
        l_off = [0]
        for i in data["LANE_IDCS"]:
            l_off.append(l_off[-1] + len(i))
        B = len(a_off) - 1
        cache = data.get("LANE_FEAT_CACHE")
        lane_shared = bool(data.get("LANE_SHARED", False))
        use_cache = cache is not None and lane_shared and cache.shape[0] * B == l_off[-1]
        want = {"actors": data["ACTORS"], "tgt_nodes": data["TGT_NODES"], "tgt_rpe": data["TGT_RPE"]}
        if "ACTOR_CTRS" in data:
            want.update(actor_ctrs=data["ACTOR_CTRS"], actor_vecs=data["ACTOR_VECS"], lane_ctrs=data["LANE_CTRS"],
                        lane_vecs=data["LANE_VECS"])
        if not use_cache:
            want["lanes"] = data["LANES"]
        out = {"a_off": a_off, "l_off": l_off, "lanes": None, "lane_feat": None, "rpe": None, "lane_shared": lane_shared,
               "actor_ctrs": None, "actor_vecs": None, "lane_ctrs": None, "lane_vecs": None}
        out.update(self._upload(want, dev))
        if "ACTOR_CTRS" not in data:   # reference-style input: precomputed RPE tensors only
            g = lambda t: (t if isinstance(t, torch.Tensor) else torch.as_tensor(t)).to(dev, torch.float32, non_blocking=True).contiguous()
            out["rpe"] = [g(r["scene"] if isinstance(r, dict) else r) for r in data["RPE"]]
        if use_cache:
            out["lane_feat"] = cache.repeat(B, 1) if B > 1 else cache
        return out

    @staticmethod
    def _upload(want, dev):
        """host arrays -> device float32 tensors through ONE packed host->device copy (every array starts on a 256-byte
        boundary of the packed buffer); tensors already on the device are passed through"""
        res, host = {}, []
        for k, v in want.items():
            if isinstance(v, torch.Tensor) and v.device.type != "cpu":
                res[k] = v.to(dev, torch.float32).contiguous()
            else:
                a = np.ascontiguousarray(v.numpy() if isinstance(v, torch.Tensor) else v, np.float32)
                host.append((k, a))
        if len(host) == 1:
            res[host[0][0]] = torch.from_numpy(host[0][1]).to(dev, non_blocking=True)
        elif host:
            offs, n = [], 0
            for _, a in host:
                offs.append(n)
                n += (a.size + 63) & ~63
            buf = np.zeros(n, np.float32)
            for (_, a), o in zip(host, offs):
                buf[o:o + a.size] = a.reshape(-1)
            dbuf = torch.from_numpy(buf).to(dev, non_blocking=True)
            for (k, a), o in zip(host, offs):
                res[k] = dbuf[o:o + a.size].view(a.shape)
        return res

    def __call__(self, d):
        if not self._loaded:
            raise RuntimeError("ScenePredNet: load_state_dict() has not been called")
        want = d["lane_feat"] is None
        o = self.rt.predict(d["actors"], d["a_off"], d["lanes"], d["l_off"], d["actor_ctrs"], d["actor_vecs"],
                            d["lane_ctrs"], d["lane_vecs"], d["tgt_nodes"], d["tgt_rpe"], rpe=d["rpe"],
                            lane_feat=d["lane_feat"], want_lane_feat=want)
        B = len(d["a_off"]) - 1
        if want and d["lane_shared"] and "lane_feat" in o:
            l0 = d["l_off"][1]
            self.last_lane_feat = o["lane_feat"][:l0]
        elif not want:
            self.last_lane_feat = d["lane_feat"][:d["l_off"][1]]
        else:
            self.last_lane_feat = None
        self.last_packed = {"n": B, "cls": o["cls"], "reg": o["reg"], "vel": o["vel"], "a_off": d["a_off"],
                            "actor_ctrs": d["actor_ctrs"], "actor_vecs": d["actor_vecs"], "rt": self.rt}
        res_cls = [o["cls"][b:b + 1] for b in range(B)]
        res_reg = [o["reg"][d["a_off"][b]:d["a_off"][b + 1]] for b in range(B)]
        res_aux = [(o["vel"][d["a_off"][b]:d["a_off"][b + 1]], None, None) for b in range(B)]
        return res_cls, res_reg, res_aux

    forward = __call__
