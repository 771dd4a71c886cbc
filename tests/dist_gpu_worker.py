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
    pl, sim, w = make_closed_loop(dict(WORKLOADS["demo1"]))
    # the sharded ranks run the round-by-round path over the host featuriser; the one-process run (the native plan) is fed by the same one,
    # so that the comparison is bit for bit (device-built root vs host featuriser: tests/test_gpu_aime_native.py)
    pl.scen_tree_gen.device_root = False
    sh = None
    if world > 1 or forced:
        sh = pl.enable_sharding()
        assert sh.world == world and sh.sharded
    res = []
    for _ in range(n_plans):
        sim.run_plans(1)
        scen, traj = sim.last_result
        res.append(dict(ctrl=np.array(sim.ctrl), best=pl.timing["best_traj_idx"], n_trees=pl.timing["n_scen_trees"],
                        keys=[list(t.nodes.keys()) for t in pl.scen_tree_gen.get_scenario_tree()],
                        xs=np.array([n.data[0] for k, n in traj[0].nodes.items() if k != -1]),
                        pos0=next(iter(scen[0].nodes.values())).data[1]))
    with open(out_path, "wb") as f:
        pickle.dump(dict(res=res, expanded=pl.scen_tree_gen.n_expanded, backend=None if sh is None else sh.backend,
                         collectives=0 if sh is None else sh.n_collectives, gathered=0 if sh is None else sh.bytes_gathered,
                         native_plans=pl.scen_tree_gen.n_native_plans), f)
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
