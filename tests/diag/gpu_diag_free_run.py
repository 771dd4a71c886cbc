"""GPU-box diagnostic: FREE-running closed loop on the recorded scenes against the reference's whole run (tests/golden/demo_runs.npz):
where does the ego state the cycles plan from start to differ?"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from bench import WORKLOADS, make_closed_loop
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
D = np.load(os.path.join(ROOT, "tests", "golden", "demo_runs.npz"))
for scene in ("demo_1", "demo_2", "demo_3", "demo_4"):
    pl, sim, w = make_closed_loop(dict(WORKLOADS[scene]), scripted=False)
    sim.episode_plans = None
    st, ci, co, keys = D[scene + "_state_in"], D[scene + "_ctrl_in"], D[scene + "_ctrl_out"], D[scene + "_scen_keys"]
    worst = 0.0; first_bad = None
    for pi in range(60):
        while not sim.step():
            pass
        d = float(np.abs(np.asarray(pl.state) - st[pi]).max())            # the state this cycle planned from
        worst = max(worst, d)
        if d > 2e-3 and first_bad is None: first_bad = (pi, d)
        k = "|".join(sim.last_result[0][0].nodes.keys())
        if k != str(keys[pi]) and first_bad is None: first_bad = (pi, "branch", k, str(keys[pi]))
    print(scene, "free-running 60 cycles: worst |state - ref| at plan time %.3e" % worst, "first deviation:", first_bad)
