"""GPU-box diagnostic: where is device work still pending during an AIME round?  (torch.cuda.synchronize() timed at probe points)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import FULL_TREE, WORKLOADS, make_closed_loop
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4tree"
pl, sim, w = make_closed_loop(dict(WORKLOADS[wl]), full_tree=wl in FULL_TREE, speculative=False)
gen = pl.scen_tree_gen
sim.run_plans(2)
def probe(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        n = len(a[0]) if a and hasattr(a[0], "__len__") else -1
        t1 = time.perf_counter(); r = f(*a, **k); d2 = (time.perf_counter() - t1) * 1e3
        print(f"  before {name}({n}): device busy {dt:.3f} ms; call {d2:.3f} ms")
        return r
    setattr(obj, name, g)
for nm in ("prune_select", "assemble_children", "get_branch_times", "update_obser_batch", "predict_scenes"):
    probe(gen, nm)
sim.run_plans(1)
