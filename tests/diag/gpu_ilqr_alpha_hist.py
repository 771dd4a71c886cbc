"""GPU-box diagnostic: which line-search candidate the tree-iLQR fits of the headline loop accept (step index 0 = alpha 1) and how the
Levenberg-Marquardt runs end -- decides what a speculative derivative pass for candidate 0 would hit."""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import BRANCHING_WEIGHTS, WORKLOADS, make_closed_loop
wl = sys.argv[1] if len(sys.argv) > 1 else "demo_1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pl, sim, w = make_closed_loop(dict(WORKLOADS[wl]), ckpt=BRANCHING_WEIGHTS)
rt, opt = pl.network.rt, pl.traj_tree_opt
opt.speculative = False
hist = [collections.Counter(), collections.Counter()]
mus = [collections.Counter(), collections.Counter()]
pred = [collections.Counter(), collections.Counter()]       # would a speculator have had the accepted candidate of slot 0, by predictor
where = [collections.Counter(), collections.Counter()]      # (slot 0 or a follower's slot, first step size or a later one) of every accepted iteration
spec = [0, 0]
nslots = int(os.environ.get("MIND_ILQR_SLOTS", "10"))
orig = opt.solve_batch


def cap(scen_trees, *a):
    r = orig(scen_trees, *a)
    for t in range(len(scen_trees)):
        for ph in (0, 1):
            tr = rt.ilqr_trace(t, ph)
            prev = 0         # step index of the fit's previous accepted iteration (0 before the first)
            run = 0          # rejections since the last accepted iteration: the slot of the accepted one is run % slots (a pass = up to `slots` iterations)
            for row in tr:
                hist[ph][int(row[2])] += 1
                if row[2] >= 0:
                    where[ph][("slot 0" if run % nslots == 0 else "slot > 0", "alpha 0" if row[2] == 0 else "alpha > 0")] += 1
                    if run % nslots == 0:
                        a = int(row[2])
                        pred[ph]["always 0"] += a == 0
                        pred[ph]["previous accepted"] += a == prev
                        pred[ph]["previous - 1"] += a == max(prev - 1, 0)
                        pred[ph]["0 or previous (two speculators)"] += a in (0, prev)
                        pred[ph]["0, previous, previous - 1 (three)"] += a in (0, prev, max(prev - 1, 0))
                        pred[ph]["accepted in slot 0"] += 1
                    prev = int(row[2])
                    run = 0
                elif row[2] == -1:
                    run += 1
                if row[2] >= 0:
                    mus[ph]["mu=0" if row[0] == 0 else ("mu<=1e-3" if row[0] <= 1e-3 else "mu>1e-3")] += 1
    a, h = rt.debug_read("il_spec")
    spec[0] += int(a); spec[1] += int(h)
    return r


opt.solve_batch = cap
sim.run_plans(n)
for ph in (0, 1):
    tot = sum(hist[ph].values())
    print("phase", ph, "accepted at", sorted(where[ph].items()))
    print("phase", ph, "speculation predictors", dict(pred[ph]))
    print("phase", ph, "iterations", tot, "accepted step index histogram", sorted(hist[ph].items()), "mu at accepted", dict(mus[ph]))
print("derivative speculator: asked in", spec[0], "passes, result adopted in", spec[1])
