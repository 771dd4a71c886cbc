"""GPU: the sharded hot path end to end.  Two ranks (gloo group, both on this box's one GPU, each with its own HIP
context) plan the same scene with every AIME round's scenes block-distributed and the contingency solves dealt
round-robin (mind_amd/parallel.py); both must plan exactly what a single process plans.  On a multi-GPU node the same
code runs one rank per GPU over RCCL (bench.py --shard)."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_gpu_worker.py")


def _run(world, tmp_path, n_plans=3, port=29531):
    procs, outs = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = os.path.join(tmp_path, f"w{world}_r{r}.pkl")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, WORKER, out, str(n_plans)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        try:
            log, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, log.decode()[-2000:]
    return [pickle.load(open(o, "rb")) for o in outs]


def test_two_sharded_ranks_plan_what_one_process_plans(tmp_path):
    single = _run(1, tmp_path)[0]
    r0, r1 = _run(2, tmp_path)
    for a, b in ((single, r0), (r0, r1)):
        for pa, pb in zip(a["res"], b["res"]):
            assert pa["keys"] == pb["keys"] and pa["best"] == pb["best"] and pa["n_trees"] == pb["n_trees"]
            assert np.array_equal(pa["pos0"], pb["pos0"])                       # predictor: bit-identical for any batch split
            assert np.array_equal(pa["xs"], pb["xs"]) and np.array_equal(pa["ctrl"], pb["ctrl"])
    assert len(single["res"][0]["keys"]) >= 2                                    # there was something to shard
    # every rank expanded only its block of each round's scenes
    assert r0["expanded"] + r1["expanded"] == single["expanded"] and 0 < r1["expanded"] < single["expanded"]


def _bench_ranks(world, extra, port):
    import json
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4",
                                       "--warmup", "1", "--backend", "gloo"] + extra, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-2000:]
        outs.append([l for l in o.splitlines() if l.strip() and not l.startswith("[Gloo]")])   # gloo's own connection notice
    assert len(outs[0]) == 1 and all(len(o) == 0 for o in outs[1:])             # rank 0 prints the ONE line
    return json.loads(outs[0][0])


def test_bench_multi_rank_contract_weak_and_strong():
    """The launch contract of `bench.py --gpus N` with N = 2 ranks (gloo here, RCCL on a multi-GPU node): barrier +
    max-over-ranks timing, whole-job value, one JSON line from rank 0.  Weak scaling: every rank plans its own scene;
    --shard: both ranks plan the same scene."""
    d = _bench_ranks(2, [], 29541)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 4 and "cpu_baseline" not in d
    assert d["config"]["sim_steps_timed"] == 20 and abs(d["value"] - 2 * 20 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    assert d["nodes_expanded_per_s"] > 0 and d["roofline"]["frac"] > 0
    s = _bench_ranks(2, ["--shard"], 29542)
    assert s["n_gpus"] == 2 and s["scaling"] == "strong"
    assert abs(s["value"] - 20 / (s["ms_per_step"] * 4e-3)) < 1e-6 * s["value"]   # one scene: steps are not multiplied by N
