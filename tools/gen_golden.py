"""Generate tests/golden/*.npz from the IMPORTED REFERENCE (build container only).

The reference Python cannot travel to the GPU box; these small fixtures (inputs are formula
generated, so only expected outputs are stored) can.  Re-run:  python tools/gen_golden.py [section ...]
Sections: predictor rpe potential ilqr aime plan
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import ref_harness as rh
from mind_amd.weights import formula_state_dict
from mind_amd.synth import predictor_batch

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
PRED_CASES = [(3, 4, 1, 1), (8, 20, 2, 1), (40, 55, 1, 1), (2, 2, 1, 3)]  # l=1 crashes the reference (Q7 .squeeze())


def to_t(pb):
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else [torch.from_numpy(x) for x in v])
            for k, v in pb.items()}


def gen_predictor():
    m = rh.ref_modules()
    sd = formula_state_dict(as_torch=True)
    net = rh.build_ref_network(sd)
    get_rpe = m["planners.mind.utils"].get_rpe
    out = {}
    for (a, l, B, seed) in PRED_CASES:
        pb = to_t(predictor_batch(a, l, B, seed=seed))
        rpes = [{"scene": get_rpe(c, v)[0], "scene_mask": None} for c, v in zip(pb["CTRS"], pb["VECS"])]
        data = (pb["ACTORS"], pb["ACTOR_IDCS"], pb["LANES"], pb["LANE_IDCS"], rpes, pb["TGT_NODES"], pb["TGT_RPE"])
        with torch.no_grad():
            act = net.actor_net(pb["ACTORS"])
            lan = net.lane_net(pb["LANES"])
        rc, rr, ra = net(data)   # autograd on, as the reference runs it (scenario_tree.py:69-71)
        key = f"a{a}_l{l}_b{B}_s{seed}"
        out[key + "_actor_net"] = act.numpy()
        out[key + "_lane_net"] = lan.numpy().reshape(-1, 128)
        out[key + "_cls"] = np.stack([c.detach().numpy()[0] for c in rc])
        reg = np.concatenate([r.detach().numpy() for r in rr])
        vel = np.concatenate([x[0].detach().numpy() for x in ra])
        if a >= 40:   # keep the fixture small: every 5th step + the last
            idx = np.r_[0:60:5, 59]
            out[key + "_tidx"] = idx
            reg, vel = reg[:, :, idx], vel[:, :, idx]
        out[key + "_reg"] = reg
        out[key + "_vel"] = vel
        out[key + "_rpe0"] = rpes[0]["scene"].numpy() if a + l <= 30 else rpes[0]["scene"].numpy()[:, :8, :8]
    np.savez_compressed(os.path.join(GOLD, "predictor.npz"), **out)
    print("predictor.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


SECTIONS = {"predictor": gen_predictor}

if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or list(SECTIONS)
    for s in which:
        SECTIONS[s]()
