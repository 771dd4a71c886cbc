"""Worker for tests/test_gpu_sharded.py: one rank of a gloo group (all ranks on the one GPU of the box) planning the SAME
scene with AIME rounds and contingency solves sharded over the ranks; writes what it planned."""
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch.distributed as dist


def main(out_path, n_plans):
    from bench import WORKLOADS, make_closed_loop
    world = int(os.environ.get("WORLD_SIZE", "1"))
    forced = os.environ.get("MIND_FORCE_COLLECTIVES", "0") == "1"     # a one-rank group whose collectives really run (parallel.Shard.force)
    if world > 1 or forced:
        backend = os.environ.get("MIND_DIST_BACKEND", "gloo")      # nccl = RCCL, one rank per GPU (LOCAL_RANK)
        if backend == "nccl":
            import torch
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
        else:
            dist.init_process_group("gloo")
    pl, sim, w = make_closed_loop(dict(WORKLOADS[os.environ.get("MIND_TEST_WORKLOAD", "demo1")]), full_tree=os.environ.get("MIND_TEST_WORKLOAD", "") in ("cfg4tree", "stress128tree"))
    # one rank and several ranks run the SAME code: mind_aime_plan, sharded through mind_set_exchange when a group is attached
    # (MIND_NATIVE_SHARD=0: the round-by-round host path of rounds 1-3 over the host featuriser, compared against the native plan fed by it)
    if os.environ.get("MIND_NATIVE_SHARD", "1") == "0":
        pl.scen_tree_gen.device_root = False
    sh = None
    if world > 1 or forced:
        sh = pl.enable_sharding()
        assert sh.world == world and sh.sharded
    res = []
    import time
    wall = []
    for _ in range(n_plans):
        t0 = time.perf_counter()
        sim.run_plans(1)
        wall.append(time.perf_counter() - t0)
        scen, traj = sim.last_result
        res.append(dict(ctrl=np.array(sim.ctrl), best=pl.timing["best_traj_idx"], n_trees=pl.timing["n_scen_trees"],
                        keys=[list(t.nodes.keys()) for t in pl.scen_tree_gen.get_scenario_tree()],
                        xs=np.array([n.data[0] for k, n in traj[0].nodes.items() if k != -1]),
                        pos0=next(iter(scen[0].nodes.values())).data[1]))
    with open(out_path, "wb") as f:
        pickle.dump(dict(res=res, expanded=pl.scen_tree_gen.n_expanded, backend=None if sh is None else sh.backend,
                         collectives=0 if sh is None else sh.n_collectives, gathered=0 if sh is None else sh.bytes_gathered,
                         native_plans=pl.scen_tree_gen.n_native_plans, wall_ms=[round(t * 1e3, 2) for t in wall], timing=dict(pl.timing_sum)), f)
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
