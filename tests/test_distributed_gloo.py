"""CPU, world_size 2 over gloo: the sharded hot path (scenes of every AIME round block-distributed over
the ranks + packed-tensor all-gather of kept children (header + world-frame rows), lane-feature broadcast; contingency solves dealt round-robin + all-gather) builds
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


def _run(world, out, port, *extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    if world == 1:
        subprocess.check_call([sys.executable, WORKER, out, *extra], env=dict(env, WORLD_SIZE="1"), timeout=300)
    else:
        subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                               "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER, out, *extra], env=env, timeout=600)
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


def test_packed_tensor_collectives_world_3(tmp_path):
    """Shard.all_gather_rows with ragged per-rank row counts (rank 0 contributes none), the round-robin gather of
    variable-size items and the broadcast, three gloo ranks: every rank ends with the same, correctly ordered tensors."""
    ranks = _run(3, str(tmp_path / "c"), 29613, "collectives")
    want_a = np.concatenate([np.arange(r * 12, dtype=np.float32).reshape(-1, 4) + 100 * r for r in range(3)])
    want_b = np.concatenate([np.full((2 + r, 2, 3), float(r), np.float32) for r in range(3)])
    sizes = [2, 1, 3, 2, 4]
    for r in ranks:
        assert np.array_equal(r["ga"], want_a) and np.array_equal(r["gb"], want_b)
        assert [x.shape[0] for x in r["rr"]] == sizes and all((x == i).all() for i, x in enumerate(r["rr"]))
        assert (r["bc"] == 7.0).all()
        assert r["n_coll"] == 3 + 2 + 1          # counts + two tensors; counts + one tensor; broadcast
        # the all-to-all with per-pair sizes (the sharded native plan's boundary-scene exchange), gloo's own and the broadcast fallback
        k = r["rank"]
        want = np.concatenate([np.full(((j + 1) * (k + 2),), 16 * j + k, np.uint8) for j in range(3) if j != k])
        assert np.array_equal(r["a2a"][0], want) and np.array_equal(r["a2a"][1], want)
        # ... and with a segment for the rank itself (forced one-rank groups send their own scenes to themselves): copied locally by the fallback
        want_self = np.concatenate([np.full(((j + 1) * (k + 2),), 16 * j + k, np.uint8) for j in range(3)])
        assert np.array_equal(r["a2a"][2], want_self) and np.array_equal(r["a2a"][3], want_self)


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
    rows = np.arange(12.0).reshape(6, 2)
    got = gather_round_robin(s, [1, 3, 2], rows)              # variable-size items, one rank: identity split
    assert [g.tolist() for g in got] == [rows[:1].tolist(), rows[1:4].tolist(), rows[4:].tolist()]
