"""GPU-box diagnostic: per-iteration traces (mu, accepted step index) of every fit of one plan of the headline loop."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import BRANCHING_WEIGHTS, WORKLOADS, make_closed_loop
wl = sys.argv[1] if len(sys.argv) > 1 else "demo_1"
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pl, sim, w = make_closed_loop(dict(WORKLOADS[wl]), ckpt=BRANCHING_WEIGHTS)
rt, opt = pl.network.rt, pl.traj_tree_opt
opt.speculative = False
sim.run_plans(skip)
orig = opt.solve_batch


def cap(scen_trees, *a):
    r = orig(scen_trees, *a)
    for t in range(len(scen_trees)):
        for ph in (0, 1):
            tr = rt.ilqr_trace(t, ph)
            print("tree", t, "M", len(scen_trees[t]._flat["parent"]) if getattr(scen_trees[t], "_flat", None) else "?", "phase", ph, " ".join("%s%.0e" % ("A%d@" % int(r_[2]) if r_[2] >= 0 else "r@", r_[0]) for r_ in tr))
    return r


opt.solve_batch = cap
sim.run_plans(2)
