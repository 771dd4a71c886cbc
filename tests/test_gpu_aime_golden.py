"""GPU: the device-side AIME glue (mind_aime_world = prune_merge's per-(agent, mode, step) arithmetic, mind_aime_rebase =
update_obser) pinned DIRECTLY against the reference's goldens (tests/golden/aime.npz: the reference ScenarioTreeGenerator
driven by the scripted FakeNet on five synthetic worlds): the same scripted modes are produced on the device, so every
round goes through k_aime_world / k_aime_rebase, and the resulting trees must carry the reference's node ids, flags,
branch times, sibling probabilities and world-frame trajectories."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class DeviceFakeNet:
    """tests/fake_net.FakeNet's scripted modes computed from the DEVICE inputs, behind ScenePredNet's device interface
    (pre_process -> device dict, __call__ leaves cls / reg / vel on the device in last_packed)."""
    computes_rpe_in_kernel = True

    def __init__(self, real, lateral=(0.0, 3.0, -3.0, 1.2, -6.0, 0.4), growth=(0.02, 0.25, 0.1, 0.02, 0.4, 0.03),
                 probs=(0.40, 0.25, 0.15, 0.10, 0.0995, 0.0005), far_mode=4):
        self.real, self.rt = real, real.rt
        self.lateral, self.growth, self.probs, self.far_mode = lateral, growth, probs, far_mode
        self.calls, self.last_lane_feat, self.last_packed = [], None, None

    def pre_process(self, data):
        d = self.real.pre_process(data)
        self._l0 = d["l_off"][1]
        return d

    def __call__(self, d):
        dev = self.rt.device
        a_off = d["a_off"]
        B = len(a_off) - 1
        self.calls.append(B)
        actors, tgt_rpe = d["actors"], d["tgt_rpe"]
        t = torch.arange(1, 61, dtype=torch.float32, device=dev) * 0.1
        cls, reg, vel = [], [], []
        for b in range(B):
            a = actors[a_off[b]:a_off[b + 1]]
            n = a.shape[0]
            idx = torch.arange(n, device=dev)
            speed = torch.sqrt(a[:, 4, -1] ** 2 + a[:, 5, -1] ** 2)
            r = torch.zeros(n, 6, 60, 5, device=dev)
            v = torch.zeros(n, 6, 60, 2, device=dev)
            for k in range(6):
                sgn = torch.where(idx % 2 == 0, 1.0, -1.0)
                lat = self.lateral[k] * sgn
                lat[0] = self.lateral[k] * (40.0 if k == self.far_mode else 0.3)
                r[:, k, :, 0] = speed[:, None] * t[None, :] * (1.0 + 0.03 * (k - 2))
                r[:, k, :, 1] = lat[:, None] * (t[None, :] / 6.0) ** 2
                sig = 0.15 + self.growth[k] * t[None, :] * (1.0 + 0.1 * idx[:, None])
                r[:, k, :, 2] = sig
                r[:, k, :, 3] = 0.8 * sig
                r[:, k, :, 4] = 1.0
                v[:, k, :, 0] = speed[:, None] * (1.0 + 0.03 * (k - 2))
                v[:, k, :, 1] = lat[:, None] * 2.0 * t[None, :] / 36.0
            p = torch.tensor(self.probs, device=dev) * (1.0 + 0.01 * torch.tanh(tgt_rpe[b].mean()))
            cls.append((p / p.sum()).view(1, 6))
            reg.append(r)
            vel.append(v)
        if d.get("lane_feat") is None:          # first round: stand-in for LaneNet's output, so that later rounds stay on the device
            self.last_lane_feat = torch.zeros(self._l0, 128, device=dev)
        else:
            self.last_lane_feat = d["lane_feat"][:d["l_off"][1]]
        self.last_packed = {"n": B, "cls": torch.cat(cls), "reg": torch.cat(reg), "vel": torch.cat(vel), "a_off": a_off,
                            "actor_ctrs": d["actor_ctrs"], "actor_vecs": d["actor_vecs"], "rt": self.rt}
        return cls, reg, [(x, None, None) for x in vel]


def test_device_aime_glue_matches_reference_goldens(hip_predictor):
    from test_aime_host import CASES, G
    from mind_amd.planners.mind.configs.planning.demo_1 import ScenTreeCfg
    from mind_amd.planners.mind.networks.network import ScenePredNet
    from mind_amd.planners.mind.planner import MINDPlanner
    from mind_amd.planners.mind.scenario_tree import DevScene, ScenarioTreeGenerator
    from mind_amd.synth import SynthWorld
    real = ScenePredNet.__new__(ScenePredNet)
    real.rt, real.device, real._loaded = hip_predictor, hip_predictor.device, True
    n_dev_rounds = 0
    for name, wkw, nkw in CASES:
        w = SynthWorld(**wkw)
        lcl = w.local_semantic_map(4.9)
        obs = w.tracks(4.9, drop={2: 30} if wkw["n_agents"] > 2 else None)
        lane, info = MINDPlanner.resample_target_lane(MINDPlanner.__new__(MINDPlanner), lcl)
        net = DeviceFakeNet(real, **nkw)
        g = ScenarioTreeGenerator(hip_predictor.device, net, 50, 50, ScenTreeCfg())
        g.reset()
        g.set_target_lane(lane, info)
        seen = []
        orig = g.predict_inputs
        g.predict_inputs = lambda scenes, orig=orig: (seen.append(all(isinstance(s, DevScene) for s in scenes)), orig(scenes))[1]
        trees = g.branch_aime(lcl, obs)
        n_dev_rounds += sum(seen[1:])
        assert list(net.calls) == list(G[name + "_batches"])
        assert list(g.tree.nodes.keys()) == list(G[name + "_internal_keys"])
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
                assert np.abs(d[1][:, ::5] - G[f"{name}_t{ti}_{k}_pos"]).max() < 1e-3       # metres, world frame
                assert np.abs(d[2][:, ::5] - G[f"{name}_t{ti}_{k}_cov"]).max() < 1e-4
                assert np.abs(np.asarray(d[3]) - G[f"{name}_t{ti}_{k}_tgt"]).max() < 1e-3
    assert n_dev_rounds >= 2          # later rounds really ran on inputs built by mind_aime_rebase
