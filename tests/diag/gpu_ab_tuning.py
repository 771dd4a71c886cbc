"""GPU-box diagnostic: same-process A/B of one mind_set_tuning knob on the headline loop (native closed loop, recorded demo_1): blocks of
plans alternate between the values on ONE context, so the box, the process and the scene history are shared and only the knob differs.
  python tests/diag/gpu_ab_tuning.py <knob> <value a> <value b> [blocks] [plans per block]
Every value must leave the results bit-identical (the knobs this is meant for do), otherwise the two arms drift apart."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import BRANCHING_WEIGHTS, WORKLOADS, make_closed_loop
knob, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 10
per = int(sys.argv[5]) if len(sys.argv) > 5 else 60          # one recorded episode (bench.make_closed_loop): every block replays the same plans
pl, sim, w = make_closed_loop(dict(WORKLOADS["demo_1"]), ckpt=BRANCHING_WEIGHTS, native=None)
assert sim._native is not None
rt = pl.network.rt
sim.run_plans(10)
tot = {va: [0.0, 0], vb: [0.0, 0]}
for b in range(2 * blocks):
    v = va if b % 2 == 0 else vb
    rt.set_tuning(knob, v)
    sim.reset()
    t0 = time.perf_counter()
    sim.run_plans(per)
    dt = time.perf_counter() - t0
    tot[v][0] += dt
    tot[v][1] += per
for v in (va, vb):
    print(f"{knob} = {v}: {tot[v][0] / tot[v][1] * 1e3:.4f} ms per plan over {tot[v][1]} plans")
print(f"difference (b - a): {(tot[vb][0] / tot[vb][1] - tot[va][0] / tot[va][1]) * 1e6:+.1f} us per plan")
