"""GPU-box diagnostic: one cfg4 full-tree plan (scripted 6-ary depth-4 AIME tree): shape of the tree and timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import WORKLOADS, make_closed_loop

if len(sys.argv) > 1 and sys.argv[1] == "trace":
    os.environ["MIND_ILQR_TRACE"] = "1"
pl, sim, w = make_closed_loop(dict(WORKLOADS["cfg4tree"]), full_tree=True)
gen = pl.scen_tree_gen
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n0 = gen.n_expanded
    sim.run_plans(1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    depth = {}
    for k, n in gen.tree.nodes.items():
        depth[n.depth] = depth.get(n.depth, 0) + 1
    trees = gen.get_scenario_tree()
    print("plan %d: %.1f ms, %d expansions, nodes per depth %s, %d scenario trees with %s nodes, timing %s" % (
        i, dt * 1e3, gen.n_expanded - n0, depth, len(trees), [len(t.nodes) for t in trees],
        {k: (round(v * 1e3, 1) if isinstance(v, float) else v) for k, v in pl.timing.items()}), flush=True)
