"""GPU-box diagnostic: closed-loop planning cycles with per-solve k_ilqr phase cycles (MIND_ILQR_TRACE)
and the host/GPU split of one planning cycle."""
import os, sys, time
os.environ["MIND_ILQR_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import WORKLOADS, make_closed_loop

wl = sys.argv[1] if len(sys.argv) > 1 else "demo1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pl, sim, w = make_closed_loop(dict(WORKLOADS[wl]))
sim.run_plans(2)
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.run_plans(1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("plan %d: cycle %.2f ms  timing %s" % (i, dt * 1e3, {k: (round(v * 1e3, 2) if isinstance(v, float) else v) for k, v in pl.timing.items()}), flush=True)
    sys.stderr.flush()
