"""CPU, world_size 2 over gloo: the sharded hot path (scenes of every AIME round block-distributed over
the ranks + all-gather of kept children; contingency solves dealt round-robin + all-gather) builds
exactly the same scenario / trajectory trees as the single-process run, on every rank."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from mind_amd.parallel import Shard, gather_round_robin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _run(world, out, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    if world == 1:
        subprocess.check_call([sys.executable, WORKER, out], env=dict(env, WORLD_SIZE="1"), timeout=300)
    else:
        subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                               "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER, out], env=env, timeout=600)
    return [pickle.load(open(f"{out}.{r}", "rb")) for r in range(world)]


def test_sharded_rounds_equal_single_process(tmp_path):
    single = _run(1, str(tmp_path / "s"), 29611)[0]
    ranks = _run(2, str(tmp_path / "d"), 29612)
    assert sum(single["calls"]) == sum(sum(r["calls"]) for r in ranks)        # same number of expansions in total
    assert len(single["calls"]) >= 2 and single["calls"][1] >= 2             # the scripted scene really branches
    for r in ranks:
        assert r["keys"] == single["keys"]                                    # identical node-id sets on every rank
        assert r["probs"] == single["probs"]
        for a, b in zip(r["pos"], single["pos"]):
            assert all(np.array_equal(x, y) for x, y in zip(a, b))           # bit-identical trajectories
        assert all(np.array_equal(x, y) for x, y in zip(r["xs"], single["xs"]))
    # the root round (1 scene) ran on rank 0 only; the scenes of the next round were split between the ranks
    assert ranks[0]["calls"][0] == 1 and len(ranks[1]["calls"]) == len(ranks[0]["calls"]) - 1
    assert ranks[0]["calls"][1] + ranks[1]["calls"][0] == single["calls"][1]
    assert ranks[0]["calls"][1] >= 1 and ranks[1]["calls"][0] >= 1


def test_block_and_round_robin_partition():
    class S(Shard):
        def __init__(self, rank, world):
            self.rank, self.world, self.active, self.group = rank, world, False, None
    for n in (0, 1, 5, 8, 13):
        for w in (1, 2, 3, 8):
            blocks = [S(r, w).block(n) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            rr = sorted(i for r in range(w) for i in S(r, w).round_robin(n))
            assert rr == list(range(n))
    s = S(0, 1)
    assert gather_round_robin(s, 3, ["a", "b", "c"]) == ["a", "b", "c"]
