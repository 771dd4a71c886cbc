"""GPU-box probe: ONE plan of the 6-ary depth-6 stress tree (128 agents x 256 lanes, plain bf16, chunked rounds): sizes, times, memory."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import WORKLOADS, make_closed_loop

budget_gb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pl, sim, w = make_closed_loop(dict(WORKLOADS["stressdeeper"]), full_tree="deeper", speculative=False)
rt = pl.network.rt
rt.set_pair_precision("bf16")
rt.set_tuning("plan_chunk_mb", budget_gb * 1024)
for i in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.run_plans(1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    info = rt.last_aime_info
    free, tot = torch.cuda.mem_get_info()
    trees = pl.scen_tree_gen.get_scenario_tree()
    scen, traj = sim.last_result
    print(f"plan {i}: {dt:.2f} s, rounds {info['round_scenes']}, aime {pl.timing['aime_s']:.2f} s, tree-iLQR {pl.timing['ilqr_s']:.2f} s, scenario-tree nodes "
          f"{sum(len(t.nodes) for t in trees)}, cost-tree nodes {[len(t.nodes) for t in traj]}, device memory in use {(tot - free) / 2**30:.1f} GiB, costs {pl.timing['tree_costs']}",
          flush=True)
