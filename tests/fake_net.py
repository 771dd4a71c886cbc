"""A scripted, deterministic stand-in for the scene predictor (tests only): K=6 modes per scene with
prescribed lateral offsets and sigma growth, a pure function of the batch inputs, so that the AIME
bookkeeping (prune / merge / branch-time / re-basing) can be pinned against the imported reference
without a trained checkpoint."""
import numpy as np
import torch


class FakeNet:
    computes_rpe_in_kernel = False

    def __init__(self, lateral=(0.0, 3.0, -3.0, 1.2, -6.0, 0.4), growth=(0.02, 0.25, 0.1, 0.02, 0.4, 0.03),
                 probs=(0.40, 0.25, 0.15, 0.10, 0.0995, 0.0005), far_mode=4):
        self.lateral = lateral
        self.growth = growth
        self.probs = probs
        self.far_mode = far_mode
        self.calls = []
        self.last_lane_feat = None

    def pre_process(self, data):
        return data

    def __call__(self, data):
        actors = data["ACTORS"] if isinstance(data, dict) else data[0]
        idcs = data["ACTOR_IDCS"] if isinstance(data, dict) else data[1]
        tgt_rpe = data["TGT_RPE"] if isinstance(data, dict) else data[6]
        actors = torch.as_tensor(actors).float().cpu()
        tgt_rpe = torch.as_tensor(tgt_rpe).float().cpu()
        self.calls.append(len(idcs))
        t = torch.arange(1, 61, dtype=torch.float32) * 0.1
        res_cls, res_reg, res_aux = [], [], []
        for b, ii in enumerate(idcs):
            a = actors[torch.as_tensor(ii).long()]
            n = a.shape[0]
            speed = torch.sqrt(a[:, 4, -1] ** 2 + a[:, 5, -1] ** 2)          # [n]
            reg = torch.zeros(n, 6, 60, 5)
            vel = torch.zeros(n, 6, 60, 2)
            for k in range(6):
                sgn = torch.where(torch.arange(n) % 2 == 0, 1.0, -1.0)
                lat = self.lateral[k] * sgn
                lat[0] = self.lateral[k] * (40.0 if k == self.far_mode else 0.3)  # ego: one mode leaves the lane
                x = speed[:, None] * t[None, :] * (1.0 + 0.03 * (k - 2))
                y = lat[:, None] * (t[None, :] / 6.0) ** 2
                reg[:, k, :, 0] = x
                reg[:, k, :, 1] = y
                sig = 0.15 + self.growth[k] * t[None, :] * (1.0 + 0.1 * torch.arange(n)[:, None])
                reg[:, k, :, 2] = sig
                reg[:, k, :, 3] = 0.8 * sig
                reg[:, k, :, 4] = 1.0
                vel[:, k, :, 0] = speed[:, None] * (1.0 + 0.03 * (k - 2))
                vel[:, k, :, 1] = lat[:, None] * 2.0 * t[None, :] / 36.0
            p = torch.tensor(self.probs) * (1.0 + 0.01 * torch.tanh(tgt_rpe[b].mean()))
            p = (p / p.sum()).view(1, 6)
            res_cls.append(p)
            res_reg.append(reg)
            res_aux.append((vel, None, None))
        return res_cls, res_reg, res_aux
