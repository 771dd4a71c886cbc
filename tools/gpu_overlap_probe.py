"""GPU-box diagnostic: does a tree-iLQR launch on one HIP context overlap with the native AIME plan of another context (same thread)?
Times scene B's plan_begin alone and while scene A's contingency solves are in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import BRANCHING_WEIGHTS, WORKLOADS, make_closed_loop
from mind_amd.pipelined import PipelinedClosedLoops

A = make_closed_loop(dict(WORKLOADS["demo_1"]), scripted=False, speculative=False, ckpt=BRANCHING_WEIGHTS, own_context=True)
B = make_closed_loop(dict(WORKLOADS["demo_1"]), scripted=False, speculative=False, ckpt=BRANCHING_WEIGHTS, own_context=True)
for pl, sim, w in (A, B):
    sim.run_plans(3)
adv = PipelinedClosedLoops._advance_to_plan
def timed0(obj, name, acc, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc.setdefault(key, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
parts = {}
for tag, L in (("A", A), ("B", B)):
    timed0(L[0].network.rt, "aime_plan", parts, tag + " aime_plan"); timed0(L[0].traj_tree_opt, "solve_batch_begin", parts, tag + " solve_batch_begin")
    timed0(L[0].scen_tree_gen, "get_scenario_tree", parts, tag + " get_scenario_tree"); timed0(L[0], "resample_target_lane", parts, tag + " resample")
    timed0(L[0].scen_tree_gen, "branch_aime", parts, tag + " branch_aime"); timed0(L[0].scen_tree_gen, "reset", parts, tag + " reset")
    timed0(L[0].scen_tree_gen, "set_target_lane", parts, tag + " set_target_lane"); timed0(L[0].traj_tree_opt, "speculate_warm", parts, tag + " speculate_warm")
    if getattr(L[0], "idle_hook", None) is not None:
        timed0(L[0], "idle_hook", parts, tag + " idle_hook")
import mind_amd.planners.mind.utils as U_
timed0(U_, "get_agent_trajectories", parts, "get_agent_trajectories"); timed0(U_, "_static_lane_pieces", parts, "_static_lane_pieces")
res = {"B alone": [], "B beside A's k_ilqr": [], "A end after B": [], "A begin": []}
for it in range(12):
    la = adv(A[1])
    t0 = time.perf_counter(); ha = A[0].plan_begin(la); res["A begin"].append(time.perf_counter() - t0)
    if it % 2 == 0:                       # B's first half while A's solves are in flight
        lb = adv(B[1])
        t0 = time.perf_counter(); hb = B[0].plan_begin(lb); res["B beside A's k_ilqr"].append(time.perf_counter() - t0)
        t0 = time.perf_counter(); ra = A[0].plan_end(ha); res["A end after B"].append(time.perf_counter() - t0)
        A[1].step_end(ra); B[1].step_end(B[0].plan_end(hb))
    else:
        A[1].step_end(A[0].plan_end(ha))
        lb = adv(B[1])
        t0 = time.perf_counter(); hb = B[0].plan_begin(lb); res["B alone"].append(time.perf_counter() - t0)
        B[1].step_end(B[0].plan_end(hb))
for k, v in list(res.items()) + sorted(parts.items()):
    print(f"{k:24s} {np.mean(v[1:]) * 1e3:.3f} ms (n = {len(v) - 1})")

