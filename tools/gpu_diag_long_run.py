"""GPU-box diagnostic: long closed-loop runs of the synthetic bench worlds (all seeds the multi-GPU / concurrent modes use),
reporting where a run stops and the ego's distance to its target lane."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import WORKLOADS, make_closed_loop, scene_workload
from mind_amd.planners.mind import utils as U
wl = sys.argv[1] if len(sys.argv) > 1 else "demo1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
for i in range(int(sys.argv[3]) if len(sys.argv) > 3 else 8):
    pl, sim, w = make_closed_loop(scene_workload(wl, i), full_tree=wl == "cfg4tree")
    lane = np.asarray(w.target_lane, dtype=np.float64)
    try:
        for p in range(n):
            sim.run_plans(1)
        msg = "ok"
    except Exception as e:
        msg = "STOPPED at plan %d: %s" % (sim.n_plans, str(e)[:80])
    d = float(np.ravel(U.get_distances_to_polyline(lane, sim.state[None, :2].astype(np.float64)))[0])
    print("scene %d: %s; t=%.2f ego x=%.1f y=%.1f v=%.2f yaw=%.2f dist to target lane %.2f m, lane x range %.0f..%.0f" % (
        i, msg, sim.sim_time, sim.state[0], sim.state[1], sim.state[2], sim.state[3], d, lane[:, 0].min(), lane[:, 0].max()))
