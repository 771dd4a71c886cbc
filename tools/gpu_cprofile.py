"""GPU-box diagnostic: cProfile of the closed loop's planning cycles (host-side cost by function)."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import FULL_TREE, WORKLOADS, make_closed_loop

wl = sys.argv[1] if len(sys.argv) > 1 else "demo_1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ckpt = sys.argv[3] if len(sys.argv) > 3 else None
pl, sim, w = make_closed_loop(dict(WORKLOADS[wl]), full_tree=wl in FULL_TREE, ckpt=ckpt)
sim.run_plans(3)
pr = cProfile.Profile()
pr.enable()
sim.run_plans(n)
pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue().replace(ROOT + "/", ""))
print("plans", n)
