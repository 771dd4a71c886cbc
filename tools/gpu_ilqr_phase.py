"""GPU-box diagnostic: per-phase cycle counters of k_ilqr on a closed-loop workload
(run with MIND_ILQR_TRACE=1 and, for the fine-grained slots, MIND_HIP_LIB=<lib built with -DIL_PROFILE>)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, make_closed_loop
wl = sys.argv[1] if len(sys.argv) > 1 else "demo_1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
os.environ.pop("MIND_ILQR_TRACE", None)
pl, sim, w = make_closed_loop(dict(WORKLOADS[wl]), ckpt=(sys.argv[3] if len(sys.argv) > 3 else None))
sim.run_plans(3)
os.environ["MIND_ILQR_TRACE"] = "1"
sim.run_plans(n)
