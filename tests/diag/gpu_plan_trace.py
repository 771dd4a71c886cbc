"""GPU-box diagnostic: MIND_PLAN_TRACE host time stamps of one mind_aime_plan call of the headline loop."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MIND_PLAN_TRACE"] = "1"
from bench import BRANCHING_WEIGHTS, WORKLOADS, make_closed_loop
pl, sim, w = make_closed_loop(dict(WORKLOADS["demo_1"]), ckpt=BRANCHING_WEIGHTS)
sim.run_plans(12)
