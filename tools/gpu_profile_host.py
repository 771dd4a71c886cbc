"""GPU-box diagnostic: cProfile of the host side of closed-loop planning cycles."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import FULL_TREE, WORKLOADS, make_closed_loop

wl = sys.argv[1] if len(sys.argv) > 1 else "demo1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pl, sim, w = make_closed_loop(dict(WORKLOADS[wl]), full_tree=wl in FULL_TREE)
sim.run_plans(3)
pr = cProfile.Profile()
pr.enable()
sim.run_plans(n)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(35)
