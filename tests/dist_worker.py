"""Worker for tests/test_distributed_gloo.py: one rank of a world_size-N gloo group running the
sharded AIME rounds (scripted FakeNet) and the round-robin contingency solves (C oracle as the
injected solver: no GPU in this container)."""
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist


def oracle_solver(cfg, flats, x0, lane, tv, use_exo, us_init):
    from oracle import ilqr as oi
    xs, us, st = [], [], []
    for i, f in enumerate(flats):
        r = oi.solve(cfg, f, x0, lane, tv, use_exo, us_init=None if us_init is None else us_init[i])
        xs.append(r["xs"]); us.append(r["us"]); st.append(dict(iterations=r["iterations"], mu=r["mu"], J=r["J"], converged=r["converged"]))
    return xs, us, st


def run(shard_on):
    from fake_net import FakeNet
    from mind_amd.parallel import Shard
    from mind_amd.planners.mind.configs.planning.demo_1 import ScenTreeCfg, TrajTreeCfg
    from mind_amd.planners.mind.planner import MINDPlanner
    from mind_amd.planners.mind.scenario_tree import ScenarioTreeGenerator
    from mind_amd.planners.mind.trajectory_tree import TrajectoryTreeOptimizer
    from mind_amd.synth import SynthWorld
    w = SynthWorld(n_agents=5, n_lanes=3, n_segs=8, seed=4)
    lcl = w.local_semantic_map(4.9)
    obs = w.tracks(4.9, drop={2: 30})
    lane, info = MINDPlanner.resample_target_lane(MINDPlanner.__new__(MINDPlanner), lcl)
    net = FakeNet(lateral=(0.0, 9.0, -9.0, 4.0, -6.0, 0.2), growth=(0.5, 0.45, 0.4, 0.5, 0.4, 0.3),
                  probs=(0.3, 0.25, 0.2, 0.15, 0.0995, 0.0005))
    g = ScenarioTreeGenerator(torch.device("cpu"), net, 50, 50, ScenTreeCfg())
    opt = TrajectoryTreeOptimizer(TrajTreeCfg())
    opt.solver = oracle_solver
    if shard_on:
        sh = Shard()
        g.shard = sh
        opt.shard = sh
    g.reset()
    g.set_target_lane(lane, info)
    trees = g.branch_aime(lcl, obs)
    tts = opt.solve_batch(trees, lcl.ego_agent.state, np.array([0.0, 0.0]), np.asarray(w.target_lane[::2], np.float64), 4.0)
    out = {"calls": list(net.calls), "keys": [list(t.nodes.keys()) for t in trees],
           "pos": [[t.nodes[k].data[1] for k in t.nodes] for t in trees],
           "probs": [[float(np.ravel(t.nodes[k].data[0])[0]) for k in t.nodes] for t in trees],
           "xs": [np.array([n.data[0] for k, n in tt.nodes.items() if k != -1]) for tt in tts]}
    return out


def collectives():
    """ragged packed all-gather (a rank with zero rows included), round-robin gather, broadcast"""
    from mind_amd.parallel import Shard, gather_round_robin
    sh = Shard()
    r, W = sh.rank, sh.world
    a = torch.arange(r * 3 * 4, dtype=torch.float32).reshape(-1, 4) + 100 * r          # rank r: 3r rows (rank 0: none)
    b = torch.full((2 + r, 2, 3), float(r))
    ga, gb = sh.all_gather_rows(a, b)
    sizes = [2, 1, 3, 2, 4]
    mine = sh.round_robin(len(sizes))
    local = np.concatenate([np.full((sizes[i], 8), float(i)) for i in mine]) if mine else np.zeros((0, 8))
    rr = gather_round_robin(sh, sizes, local)
    t = torch.full((5,), float(r + 7), device=sh.device)
    sh.broadcast(t, 0)
    n_coll = sh.n_collectives
    # the all-to-all of the sharded native plan's exchange (MIND_XCHG_ALLTOALLV) as its CPU transport performs it: rank j sends (j + 1) * (k + 2)
    # bytes of value 16 j + k to rank k != j, nothing to itself -- through gloo's own all-to-all where this build has it, and through the
    # broadcast fallback
    # ... and once more with a segment for the rank itself (a forced one-rank group sends its own re-based scenes to itself): the fallback must
    # copy it locally, into an output buffer that was NOT zeroed
    a2a = []
    for force_fallback, self_too in ((False, False), (True, False), (False, True), (True, True)):
        sb = [0 if (k == r and not self_too) else (r + 1) * (k + 2) for k in range(W)]
        rb = [0 if (j == r and not self_too) else (j + 1) * (r + 2) for j in range(W)]
        inp = torch.cat([torch.full((sb[k],), 16 * r + k, dtype=torch.uint8) for k in range(W)]) if sum(sb) else torch.empty(0, dtype=torch.uint8)
        out = torch.full((sum(rb),), 255, dtype=torch.uint8)
        if force_fallback:
            real = dist.all_to_all_single

            def refuse(*a_, **k_):
                raise RuntimeError("no all-to-all in this build")
            dist.all_to_all_single = refuse
            try:
                sh._all_to_all_cpu(out, inp, rb, sb)
            finally:
                dist.all_to_all_single = real
        else:
            sh._all_to_all_cpu(out, inp, rb, sb)
        a2a.append(out.numpy().copy())
    return {"ga": ga.numpy(), "gb": gb.numpy(), "rr": [x.copy() for x in rr], "bc": t.numpy(), "n_coll": n_coll, "a2a": a2a, "rank": r}


if __name__ == "__main__":
    out_path = sys.argv[1]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    res = collectives() if len(sys.argv) > 2 and sys.argv[2] == "collectives" else run(world > 1)
    rank = dist.get_rank() if world > 1 else 0
    with open(f"{out_path}.{rank}", "wb") as f:
        pickle.dump(res, f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
