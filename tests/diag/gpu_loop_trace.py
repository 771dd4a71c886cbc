"""GPU-box diagnostic: MIND_LOOP_TRACE + MIND_PLAN_TRACE host time stamps of a few native planning cycles of the headline loop."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MIND_LOOP_TRACE"] = "1"
os.environ["MIND_PLAN_TRACE"] = "1"
from bench import BRANCHING_WEIGHTS, WORKLOADS, make_closed_loop
pl, sim, w = make_closed_loop(dict(WORKLOADS["demo_1"]), ckpt=BRANCHING_WEIGHTS, native=None)
assert sim._native is not None
sim.run_plans(14)
