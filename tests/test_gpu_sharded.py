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


def _run(world, tmp_path, n_plans=3, port=29531, extra_env=None):
    procs, outs = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
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
    # one code path for every rank count: the native plan (mind_aime_plan), its rounds sharded through mind_set_exchange
    assert single["native_plans"] == 3 and r0["native_plans"] == 3 and r1["native_plans"] == 3 and r0["collectives"] >= 3 * 4


def test_three_ranks_and_the_full_tree_sharded_natively(tmp_path):
    """The full scripted 6-ary tree of BASELINE configs[3] (rounds of 1 / 6 / 36 / 216 scenes, six cost trees) through the sharded native
    plan on three ranks (ragged blocks: 216 = 72 x 3, 36 = 12 x 3, 6 = 2 x 3, the root round on rank 0 alone): every rank returns the plan the
    single process returns, bit for bit, and expands its share of the 259 scenes."""
    env = {"MIND_TEST_WORKLOAD": "cfg4tree"}
    single = _run(1, tmp_path, n_plans=1, port=29571, extra_env=env)[0]
    ranks = _run(3, tmp_path, n_plans=1, port=29572, extra_env=env)
    assert single["expanded"] == 259 and sum(r["expanded"] for r in ranks) == 259 and [r["expanded"] for r in ranks] == [87, 86, 86]
    print("cfg4tree on 3 ranks: collectives per rank", [r["collectives"] for r in ranks], "MB received per rank", [round(r["gathered"] / 1e6, 2) for r in ranks])
    for b in ranks:
        assert b["native_plans"] == 1
        # what a rank receives per plan (mind_last_exchange_stats): the decisions and frames of every scene, LaneNet's output once, the
        # BOUNDARY scenes of the next round's blocks (all-to-all; round 4's all-gather handed every rank all 258 re-based scenes: 66 MB),
        # the completed rows / cost trees
        assert b["gathered"] < 10e6, b["gathered"]
        for pa, pb in zip(single["res"], b["res"]):
            assert pa["keys"] == pb["keys"] and pa["best"] == pb["best"] and pa["n_trees"] == pb["n_trees"] == 6
            assert np.array_equal(pa["pos0"], pb["pos0"]) and np.array_equal(pa["xs"], pb["xs"]) and np.array_equal(pa["ctrl"], pb["ctrl"])


def test_sharded_rounds_in_chunks_plan_what_one_process_plans_whole(tmp_path):
    """Sharding and chunking together: two ranks, the full cfg4 tree, every rank's block of a round through the predictor in chunks under
    a 2 GB edge budget (the 108-scene blocks of the last round in three chunks of 37) -- the plan of the single process that ran every
    round whole, bit for bit."""
    single = _run(1, tmp_path, n_plans=1, port=29591, extra_env={"MIND_TEST_WORKLOAD": "cfg4tree"})[0]
    ranks = _run(2, tmp_path, n_plans=1, port=29592, extra_env={"MIND_TEST_WORKLOAD": "cfg4tree", "MIND_PLAN_CHUNK_MB": "2048"})
    assert single["expanded"] == 259 and [r["expanded"] for r in ranks] == [130, 129]
    for b in ranks:
        assert b["native_plans"] == 1
        for pa, pb in zip(single["res"], b["res"]):
            assert pa["keys"] == pb["keys"] and pa["best"] == pb["best"] and pa["n_trees"] == pb["n_trees"] == 6
            assert np.array_equal(pa["pos0"], pb["pos0"]) and np.array_equal(pa["xs"], pb["xs"]) and np.array_equal(pa["ctrl"], pb["ctrl"])


def test_round_by_round_host_path_still_shards(tmp_path):
    """MIND_NATIVE_SHARD=0: the sharded rounds of rounds 1-3 (round-by-round host path over parallel.Shard.all_gather_rows, kept for
    wrapped networks) against the native one-process plan fed by the same host featuriser."""
    env = {"MIND_NATIVE_SHARD": "0"}
    single = _run(1, tmp_path, port=29581, extra_env=env)[0]
    r0, r1 = _run(2, tmp_path, port=29582, extra_env=env)
    for a, b in ((single, r0), (r0, r1)):
        for pa, pb in zip(a["res"], b["res"]):
            assert pa["keys"] == pb["keys"] and pa["best"] == pb["best"] and np.array_equal(pa["pos0"], pb["pos0"]) and np.array_equal(pa["xs"], pb["xs"])
    assert r0["native_plans"] == 0 and r0["expanded"] + r1["expanded"] == single["expanded"]


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
    max-over-ranks timing, whole-job value, one JSON line from rank 0.  Weak scaling: every rank runs its own closed loop;
    --shard: both ranks plan the same scene."""
    d = _bench_ranks(2, ["--workload", "demo1", "--no-extras"], 29541)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 4 and "cpu_baseline" not in d
    assert d["config"]["sim_steps_timed"] == 20 and abs(d["value"] - 2 * 20 / (d["ms_per_step"] * 4e-3)) < 1e-3 * d["value"]      # (the line rounds to 5 significant digits)
    assert d["nodes_expanded_per_s"] > 0 and 0 < d["roofline"]["frac"] <= 1
    s = _bench_ranks(2, ["--workload", "demo1", "--shard"], 29542)
    assert s["n_gpus"] == 2 and s["scaling"] == "strong" and s["collectives_per_plan"] >= 3
    assert abs(s["value"] - 20 / (s["ms_per_step"] * 4e-3)) < 1e-3 * s["value"]   # one scene: steps are not multiplied by N


def test_bench_spawns_its_own_ranks_and_checks_the_world_size(tmp_path):
    """`python bench.py --gpus 2` with no launcher in the environment starts the two ranks itself (torch.distributed.run on
    127.0.0.1) and reports n_gpus = 2; the default multi-rank line carries the sharded cfg4-tree plan next to the replicas;
    a launcher world that disagrees with --gpus is refused."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["MIND_BENCH_EXTRAS"] = str(tmp_path / "bench_extras.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
                          "--tree-steps", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert len(lines[0]) < 4096
    c = json.loads(lines[0])
    assert c["n_gpus"] == 2 and c["scaling"] == "weak" and "recorded scene demo_1" in c["config"]["workload"]
    # the compact line carries north_star's strong-scaling figures of the sharded tree; the blocks in full are in the extras file
    assert c["tree_sharded"]["nodes_expanded_per_s"] > 0 and c["tree_sharded"]["speedup_vs_1"] > 0 and c["tree_sharded"]["gathered_mb_per_plan"] > 0
    assert c["tree_replicas"]["nodes_expanded_per_s"] > 0
    d = json.load(open(env["MIND_BENCH_EXTRAS"]))
    t = d["tree_sharded"]
    assert t["n_gpus"] == 2 and t["scaling"] == "strong" and t["expansions_per_plan"] == 259 and t["collectives_per_plan"] >= 9
    w = d["tree_replicas"]          # the same tree, one independent scene per rank: twice the expansions in the same plan time
    assert w["n_gpus"] == 2 and w["scaling"] == "weak" and w["expansions_per_plan"] == 2 * 259 and "collectives_per_plan" not in w
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr


def test_two_ranks_over_rccl_when_two_devices_are_visible(tmp_path):
    """The same sharded plan over the nccl (= RCCL) backend, one rank per GPU: needs two devices, skipped on the one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL runs one rank per device)")
    single = _run(1, tmp_path)[0]
    procs, outs = [], []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT="29551",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", MIND_DIST_BACKEND="nccl")
        out = os.path.join(tmp_path, f"nccl_r{r}.pkl")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, WORKER, out, "3"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        log, _ = p.communicate(timeout=300)
        assert p.returncode == 0, log.decode()[-2000:]
    for o in outs:
        b = pickle.load(open(o, "rb"))
        for pa, pb in zip(single["res"], b["res"]):
            assert pa["keys"] == pb["keys"] and pa["best"] == pb["best"]
            assert np.array_equal(pa["pos0"], pb["pos0"]) and np.array_equal(pa["xs"], pb["xs"])


def test_one_rank_rccl_group_runs_every_collective_of_the_sharded_plan(tmp_path):
    """RCCL executed on the one-GPU box: a world-size-1 `nccl` process group with the short-cuts of parallel.Shard switched off
    (MIND_FORCE_COLLECTIVES=1), so every exchange step of the sharded plan -- the per-round packed all_gather_into_tensor of the
    kept children (header + world-frame rows, device tensors), the broadcast of LaneNet's output, the round-robin gather of the
    tree-iLQR rows -- goes through the RCCL backend.  The plan must equal the plain single-process plan bit for bit."""
    single = _run(1, tmp_path)[0]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561",
               HSA_ENABLE_IPC_MODE_LEGACY="0", MIND_DIST_BACKEND="nccl", MIND_FORCE_COLLECTIVES="1")
    out = os.path.join(tmp_path, "nccl_w1.pkl")
    p = subprocess.run([sys.executable, WORKER, out, "3"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    b = pickle.load(open(out, "rb"))
    assert b["backend"] == "nccl" and b["collectives"] >= 3 * 4 and b["gathered"] > 0          # >= 3 plans x (rounds + broadcast + solves)
    for pa, pb in zip(single["res"], b["res"]):
        assert pa["keys"] == pb["keys"] and pa["best"] == pb["best"] and pa["n_trees"] == pb["n_trees"]
        assert np.array_equal(pa["pos0"], pb["pos0"]) and np.array_equal(pa["xs"], pb["xs"]) and np.array_equal(pa["ctrl"], pb["ctrl"])
    assert b["expanded"] == single["expanded"]
