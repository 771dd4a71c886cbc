"""GPU-box diagnostic: HIP tree-iLQR vs the C oracle on scripted scenario trees."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from mind_amd.predictor import HipPredictor
from mind_amd.synth import scripted_scenario_tree
from oracle import ilqr as oi


def main():
    hp = HipPredictor(0)
    ok = True
    cases = [("straight", 3), ("lead", 4), ("branch3", 6), ("deep", 3), ("branch3", 40)]
    for kind, a in cases:
        sst = scripted_scenario_tree(kind, a)
        for max_iter in ([1, 2, 4, 100] if kind != "deep" else [1, 2, 4]):
            cfg = oi.default_cfg(max_iter=max_iter)
            flat = oi.flatten(sst["nodes"])
            x0 = oi.init_state(sst["state"], sst["ctrl"])
            t0 = time.time()
            w = oi.solve(cfg, flat, x0, sst["target_lane"], sst["target_vel"], 0)
            f = oi.solve(cfg, flat, x0, sst["target_lane"], sst["target_vel"], 1, us_init=w["us"])
            t1 = time.time()
            xs_w, us_w, st_w = hp.ilqr_solve(cfg, [flat], x0, sst["target_lane"], sst["target_vel"], 0)
            xs_f, us_f, st_f = hp.ilqr_solve(cfg, [flat], x0, sst["target_lane"], sst["target_vel"], 1, us_init=[w["us"]])
            t2 = time.time()
            dw = np.abs(xs_w[0] - w["xs"]).max()
            df = np.abs(xs_f[0] - f["xs"]).max()
            print(f"{kind:9s} a={a:2d} M={len(flat['parent']):3d} it<={max_iter:3d}: warm dx {dw:.2e} (it {st_w[0]['iterations']}/{w['iterations']} J {st_w[0]['J']:.9g}/{w['J']:.9g})"
                  f" full dx {df:.2e} du {np.abs(us_f[0]-f['us']).max():.2e} (it {st_f[0]['iterations']}/{f['iterations']} J {st_f[0]['J']:.9g}/{f['J']:.9g} mu {st_f[0]['mu']:g}/{f['mu']:g})"
                  f" t_oracle {t1-t0:.3f}s t_hip {t2-t1:.4f}s")
            if not (dw < 1e-6 and df < 1e-6):
                ok = False
    # multi-tree call + timing
    sst = scripted_scenario_tree("branch3", 40)
    flat = oi.flatten(sst["nodes"])
    cfg = oi.default_cfg()
    x0 = oi.init_state(sst["state"], sst["ctrl"])
    for n in (1, 6):
        hp.ilqr_solve(cfg, [flat] * n, x0, sst["target_lane"], sst["target_vel"], 1)
        t0 = time.time()
        for _ in range(5):
            xs, us, st = hp.ilqr_solve(cfg, [flat] * n, x0, sst["target_lane"], sst["target_vel"], 1)
        dt = (time.time() - t0) / 5
        print(f"timing {n} trees x M={len(flat['parent'])} a=40 full solve from zero: {dt*1e3:.2f} ms, iterations {[s['iterations'] for s in st]}")
    print("DIAG_OK" if ok else "DIAG_FAIL")


if __name__ == "__main__":
    main()
