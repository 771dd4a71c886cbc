"""GPU-box diagnostic: wall time of the first planning cycles of a fresh process (where does the first measurement's slowness sit?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import BRANCHING_WEIGHTS, WORKLOADS, make_closed_loop
torch.cuda.set_stream(torch.cuda.Stream())
for rep in range(2):
    pl, sim, w = make_closed_loop(dict(WORKLOADS["demo_1"]), ckpt=BRANCHING_WEIGHTS)
    ts = []
    for i in range(48):
        t0 = time.perf_counter(); sim.run_plans(1); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("planner", rep, "ms per cycle:", " ".join("%.2f" % t for t in ts), "| aime", " ".join("%.2f" % 0 for _ in []))
    print("  mean of cycles 3-22: %.3f, 23-42: %.3f" % (sum(ts[3:23]) / 20, sum(ts[23:43]) / 20), "spec counters", dict(pl.traj_tree_opt.counters))
