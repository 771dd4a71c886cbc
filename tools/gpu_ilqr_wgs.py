"""GPU-box diagnostic: tree-iLQR wall time per plan on a full-tree workload as a function of the workgroups per cost tree."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import FULL_TREE, WORKLOADS, make_closed_loop
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4tree"
pl, sim, w = make_closed_loop(dict(WORKLOADS[wl]), full_tree=wl in FULL_TREE)
rt = pl.network.rt
sim.run_plans(1)
for G in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8,16,32").split(",")]:
    rt.set_tuning("ilqr_wgs", G)
    sim.run_plans(1)
    t0 = dict(pl.timing_sum)
    sim.run_plans(3)
    t1 = pl.timing_sum
    n = t1["plans"] - t0["plans"]
    print(f"{wl}: {G:2d} workgroups per tree: tree-iLQR {1e3 * (t1['ilqr_s'] - t0['ilqr_s']) / n:7.2f} ms per plan, plan {1e3 * (t1['total_s'] - t0['total_s']) / n:7.2f} ms")
