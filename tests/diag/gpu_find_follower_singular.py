"""GPU-box diagnostic: search control weights w = -mu* dt^2 / 2 (mu* a power of two) for which a fit meets its singular Q_uu right behind a rejection,
i.e. in a follower's slot (tests/test_gpu_ilqr.py::test_singular_q_uu_in_a_followers_slot)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from mind_amd.predictor import HipPredictor
from mind_amd.synth import scripted_scenario_tree
from oracle import ilqr as oi
hp = HipPredictor(0)
hp.set_tuning("ilqr_slots", 1)
for kind, n in (("straight", 4), ("lead", 4), ("branch3", 6), ("deep", 5)):
    sst = scripted_scenario_tree(kind, n)
    flat = oi.flatten(sst["nodes"]); x0 = oi.init_state(sst["state"], sst["ctrl"])
    for exo in (0, 1):
        for e in range(-12, 8):
            cfg = oi.default_cfg(max_iter=30)
            w = -(2.0 ** e) * 0.2 * 0.2 / 2
            cfg.w_ctrl[:] = [w, w]
            hp.ilqr_solve(cfg, [flat], x0, sst["target_lane"], sst["target_vel"], exo)
            tr = hp.ilqr_trace(0, 0)
            p = tr[:, 2]
            hit = [i for i in range(1, len(p)) if p[i] == -2 and p[i - 1] == -1]
            if hit:
                print(kind, n, "exo", exo, "mu*=2^%d" % e, "rows", len(p), "first singular-behind-rejection at", hit[0], "picks", p.astype(int).tolist())
