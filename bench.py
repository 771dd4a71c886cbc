#!/usr/bin/env python
"""bench.py --gpus N --steps K --warmup W [--workload demo1|cfg4]

A step = one planning cycle of the closed loop on one synthetic scene (mind_amd.closed_loop, mirroring
simulator.py:58-103 / agent.py:277-331): 5 simulator steps of 0.02 s (observation fan-out, 10 Hz planner
trigger, ego plant) containing one MINDPlanner.plan() = AIME scenario tree (every tree node through the
HIP predictor) + tree-iLQR contingency solves (warm start + full) for every scenario tree + selection.
value = simulator steps / wall time.  One process per GPU; ranks run independent scenes (weak scaling, no
data-path collective); rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np
import torch

WORKLOADS = {
    # demo_1-like: ~40 tracked agents, ~55 lane polylines (a<=~40, l~55, N~96)
    "demo1": dict(n_agents=40, n_lanes=5, n_segs=11, seed=3),
    # cfg4: 64 agents x 256 lane polylines
    "cfg4": dict(n_agents=64, n_lanes=8, n_segs=32, seed=4),
    # cfg4 with the full scripted 6-ary depth-4 AIME tree (259 expansions / plan; BASELINE config 4)
    "cfg4tree": dict(n_agents=64, n_lanes=8, n_segs=32, seed=4),
    # the agent count of BASELINE config 5 (128 agents x 256 lane polylines, N = 385) on the largest tree the
    # reference's probability floor lets grow (6-ary, 259 expansions), fp32
    "stress128tree": dict(n_agents=128, n_lanes=8, n_segs=32, seed=5),
    # the reference's four recorded AV2 demo scenes (compact fixtures derived from data/<seq_id>/, tests/golden/scenes)
    "demo_1": dict(scene="demo_1"), "demo_2": dict(scene="demo_2"), "demo_3": dict(scene="demo_3"), "demo_4": dict(scene="demo_4"),
    # BASELINE config 3: demo_{1,2,3,4} concurrently on one GPU (use with --concurrent P: scene i plans demo_(i mod 4 + 1))
    "demo_all": dict(scene="demo_1"),
}


def scene_workload(workload, i):
    """workload of the i-th concurrent scene: its own seed for the synthetic worlds, round-robin over the four recorded scenes
    for demo_all."""
    if workload == "demo_all":
        return dict(scene="demo_%d" % (i % 4 + 1))
    wkw = dict(WORKLOADS[workload])
    if "seed" in wkw:
        wkw["seed"] = wkw["seed"] + i
    return wkw


def _concurrent_label(workload, P):
    if workload == "demo_all":
        return f"the recorded scenes demo_1..4 ({P} closed loops, scene i = demo_(i mod 4 + 1))", "recorded AV2 scenes, formula-initialised weights"
    if "scene" in WORKLOADS[workload]:
        return f"{P} closed loops on the recorded scene {workload}", "recorded AV2 scene, formula-initialised weights"
    return f"{P} {workload}-like synthetic scenes", "synthetic"
FULL_TREE = ("cfg4tree", "stress128tree")
F_MIN_N2 = 754944.0   # minimal-algorithm FLOPs per expansion, N^2 coefficient (6 layers)
PEAK_F32_MFMA = 157.3e12


def make_planner(wkw, scripted=True):
    from mind_amd.planners.mind.planner import MINDPlanner
    from mind_amd.synth import ScriptedBranching, SynthWorld
    cfg = os.path.join(ROOT, "mind_amd", "planners", "mind", "configs", "synthetic.json")
    w = SynthWorld(**wkw)
    pl = MINDPlanner(cfg)
    if scripted:
        pl.scen_tree_gen.network = ScriptedBranching(pl.network)
    for s in range(50):
        pl.update_observation(w.local_semantic_map(round(0.1 * s, 6)))
    lcl = w.local_semantic_map(4.9)
    pl.update_target_lane(np.asarray(w.target_lane[::2], dtype=np.float64))
    pl.update_state_ctrl(lcl.ego_agent.state, np.array([0.0, 0.0]))
    return pl, lcl, w


def make_closed_loop(wkw, scripted=True, full_tree=False, speculative=True):
    """planner + closed-loop simulator advanced to the enable time (t = 4.0 s: 40 observation updates)."""
    from mind_amd.closed_loop import ClosedLoopSim
    from mind_amd.planners.mind.planner import MINDPlanner
    from mind_amd.synth import ScriptedBranching, ScriptedFullTree, SynthWorld
    cfg = os.path.join(ROOT, "mind_amd", "planners", "mind", "configs", "synthetic.json")
    if "scene" in wkw:
        from mind_amd.scene_io import ReplayWorld, scene_fixture_path
        w = ReplayWorld.from_scene_file(scene_fixture_path(wkw["scene"]))
        cfg = dict(json.load(open(cfg)), planning_config="planners.mind.configs.planning." + wkw["scene"])
    else:
        w = SynthWorld(**wkw)
    pl = MINDPlanner(cfg)
    # recorded scenes run the predictor's own modes, as the reference does with the same weights (its AIME tree then
    # collapses to a few nodes); the scripted modes are straight-line motions in the agent frame and would leave a
    # curved recorded target lane, so they are kept for the synthetic worlds only
    if scripted and "scene" not in wkw:
        pl.scen_tree_gen.network = (ScriptedFullTree if full_tree else ScriptedBranching)(pl.network)
    # episodes: the reference's 60 cycles for a recording; 24 for the synthetic worlds (all eight seeds the weak-scaling
    # and concurrent modes use stay on their lane that long; the default 3 + 20 cycles fit in one episode)
    # speculative warm start = a second HIP context per planner: a latency lever for a GPU that one closed loop leaves idle;
    # with many scenes sharing the device the extra contexts cost more than they hide (measured: 8 processes 3150 -> 2010)
    pl.traj_tree_opt.speculative = speculative and pl.traj_tree_opt.speculative      # MIND_SPECULATIVE_WARM_START=0 switches it off
    sim = ClosedLoopSim(w, pl, episode_plans=60 if "scene" in wkw else 24)
    sim.run_until(sim.enable_time)
    return pl, sim, w


def recorded_scenes(plans=20, warmup=3):
    """The same closed loop on the reference's four recorded AV2 demo scenes (compact fixtures, tests/golden/scenes) with
    the predictor's own modes: K timed planning cycles each, synchronised on both sides.  Reported next to the headline
    (whose synthetic demo_1-like scene adds the branching a trained checkpoint would produce)."""
    out = {}
    for name in ("demo_1", "demo_2", "demo_3", "demo_4"):
        pl, sim, w = make_closed_loop(dict(WORKLOADS[name]))
        sim.run_plans(warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n0 = pl.scen_tree_gen.n_expanded
        steps = sim.run_plans(plans)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"sim_steps_per_s": steps / dt, "ms_per_plan": dt / plans * 1e3, "agents": len(pl.agent_obs),
                     "lane_polylines": int(pl.scen_tree_gen.lane_feat_in.shape[0]),
                     "expansions_per_plan": (pl.scen_tree_gen.n_expanded - n0) / plans}
        if name == "demo_1":
            # BASELINE configs[0]/[1]: the whole demo_1 closed loop = 500 simulator steps (10 s), 60 planning cycles
            from mind_amd.closed_loop import ClosedLoopSim
            from mind_amd.planners.mind.planner import MINDPlanner
            sim = ClosedLoopSim(w, MINDPlanner(pl.planner_cfg))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(500):
                sim.step()
            torch.cuda.synchronize()
            out[name]["whole_run_500_steps_s"] = time.perf_counter() - t0
            out[name]["whole_run_plans"] = sim.n_plans
    out["note"] = ("recorded map + tracks, formula-initialised weights (the trained checkpoint is not in the reference tree); parity of "
                   "this loop against the reference's own simulator: tests/test_gpu_plan.py::test_recorded_demo_scenes_match_reference_closed_loop")
    return out


def cpu_baseline(pl, lcl, n_scene_tokens, expansions, scen_trees):
    """The oracle (CPU restatement, kind 'port') on this host: one predictor forward of the same scene
    size + the contingency solves of this plan's scenario trees, timed on a bounded sample."""
    from mind_amd.synth import predictor_batch
    from mind_amd.weights import formula_state_dict
    from oracle import ilqr as oi
    from oracle import predictor as op
    sd = formula_state_dict(as_torch=True)
    a, l = n_scene_tokens
    pb = predictor_batch(a, l, 1, seed=9)
    tb = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else [torch.from_numpy(x) for x in v]) for k, v in pb.items()}
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 8.0 and reps < 20:
        op.forward(sd, tb)
        reps += 1
    t_pred = (time.perf_counter() - t0) / reps
    cfg = oi.default_cfg()
    t0 = time.perf_counter()
    n_tree = 0
    for st in scen_trees[:3]:
        nodes = [(k, n.parent_key, n.data) for k, n in st.nodes.items()]
        oi.contingency(cfg, nodes, pl.state, pl.ctrl, pl.gt_tgt_lane, lcl.target_velocity)
        n_tree += 1
    t_ilqr = (time.perf_counter() - t0) / max(n_tree, 1)
    plan_s = expansions * t_pred + len(scen_trees) * t_ilqr
    return {"value": 5.0 / plan_s, "unit": "sim steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} oracle predictor forwards (a={a}, l={l}; {t_pred*1e3:.0f} ms each, torch-CPU fp32) + "
                      f"{n_tree} oracle C tree-iLQR contingency solves with materialised 256x256 fields "
                      f"({t_ilqr*1e3:.0f} ms each, 1 thread); plan = {expansions} expansions + {len(scen_trees)} solves",
            "plan_ms": plan_s * 1e3}


def _proc_scene(i, workload, steps, warmup, ready, go, q, speculative):
    """One scene in its own process (own HIP context): signals ready, waits for the common start, reports back."""
    import torch as th
    pl, sim, w = make_closed_loop(scene_workload(workload, i), full_tree=workload in FULL_TREE, speculative=speculative)
    sim.run_plans(max(warmup, 1))
    th.cuda.synchronize()
    ready.wait()
    go.wait()
    t0 = time.time()
    n = sim.run_plans(steps)
    th.cuda.synchronize()
    q.put((i, n, t0, time.time(), pl.scen_tree_gen.n_expanded))


def run_concurrent_processes(args):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    P = args.concurrent
    ready, go, q = ctx.Barrier(P + 1), ctx.Barrier(P + 1), ctx.Queue()
    procs = [ctx.Process(target=_proc_scene, args=(i, args.workload, args.steps, args.warmup, ready, go, q, P <= 4)) for i in range(P)]
    for p_ in procs:
        p_.start()
    ready.wait(timeout=600)
    go.wait(timeout=60)
    res = [q.get(timeout=600) for _ in range(P)]
    for p_ in procs:
        p_.join(timeout=60)
    dt = max(r[3] for r in res) - min(r[2] for r in res)       # same host clock: first start to last finish
    steps = sum(r[1] for r in res)
    print(json.dumps({
        "metric": "sim steps/sec (whole node) + scenario-tree nodes expanded/sec, AV2 demo scenes",
        "value": steps / dt, "unit": "sim steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 predictor / f64 iLQR", "data": _concurrent_label(args.workload, P)[1],
        "config": {"workload": f"{_concurrent_label(args.workload, P)[0]} planned concurrently on one GPU (one host process + HIP "
                               f"context per scene), {args.steps} planning cycles each", "concurrent_scenes": P,
                   "sim_steps_timed": steps},
        "ms_per_plan_aggregate": dt / (args.steps * P) * 1e3}))


def run_concurrent(args):
    """P closed loops (different scenes) in P host threads, each with its own HIP context on its own stream: the
    GPU work of one scene overlaps the host bookkeeping and the GPU work of the others."""
    import threading
    P = args.concurrent
    loops, errs = [None] * P, []
    ready, go = threading.Barrier(P + 1), threading.Barrier(P + 1)
    done_steps = [0] * P

    def worker(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                pl, sim, w = make_closed_loop(scene_workload(args.workload, i), full_tree=args.workload in FULL_TREE, speculative=P <= 4)
                sim.run_plans(max(args.warmup, 1))
                torch.cuda.current_stream().synchronize()
                loops[i] = (pl, sim)
                ready.wait()
                go.wait()
                done_steps[i] = sim.run_plans(args.steps)
                torch.cuda.current_stream().synchronize()
        except Exception as e:           # surface the failure instead of dead-locking the barriers
            errs.append(e)
            for b in (ready, go):
                b.abort()

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(P)]
    for t in ths:
        t.start()
    ready.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go.wait()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if errs:
        raise errs[0]
    steps = sum(done_steps)
    exp = sum(pl.scen_tree_gen.n_expanded for pl, _ in loops)
    print(json.dumps({
        "metric": "sim steps/sec (whole node) + scenario-tree nodes expanded/sec, AV2 demo scenes",
        "value": steps / dt, "unit": "sim steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 predictor / f64 iLQR", "data": _concurrent_label(args.workload, P)[1],
        "config": {"workload": f"{_concurrent_label(args.workload, P)[0]} planned concurrently on one GPU (one host thread + HIP "
                               f"context + stream per scene), {args.steps} planning cycles each", "concurrent_scenes": P,
                   "sim_steps_timed": steps},
        "ms_per_plan_aggregate": dt / (args.steps * P) * 1e3}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="demo1", choices=list(WORKLOADS))
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for --gpus N > 1 (nccl = RCCL; gloo only for tests on a single-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recorded", action="store_true", help="skip the extra closed loops on the four recorded demo scenes")
    ap.add_argument("--concurrent", type=int, default=0,
                    help="BASELINE config 3: plan this many independent scenes concurrently on the GPU (one host thread, "
                         "HIP context and stream per scene); prints the aggregate rate")
    ap.add_argument("--processes", action="store_true",
                    help="with --concurrent: one host PROCESS per scene instead of one thread (host bookkeeping in parallel too)")
    ap.add_argument("--shard", action="store_true",
                    help="strong scaling: all ranks plan the SAME scene, AIME rounds and contingency solves sharded over ranks")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":       # RCCL over xGMI: one rank per GPU
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:                            # gloo: the same code path on a box with fewer GPUs than ranks (tests)
            dist.init_process_group(args.backend)
    if args.concurrent > 1:
        return run_concurrent_processes(args) if args.processes else run_concurrent(args)
    wkw = dict(WORKLOADS[args.workload])
    if not args.shard and "seed" in wkw:
        wkw["seed"] = wkw["seed"] + rank      # every rank plans its own scene (weak scaling)
    real_scene = "scene" in wkw
    pl, sim, w = make_closed_loop(wkw, full_tree=args.workload in FULL_TREE)
    if args.shard and dist is not None:
        pl.enable_sharding()
    rt = pl.network.rt
    sim.run_plans(max(args.warmup, 1))
    rt.set_profiling(True)                     # HIP-event timing of the fusion pair kernels on the ctx stream
    pair_ms, pair_launch, pair_n2 = [], 0, 0.0
    pair_exec = [0.0]
    expansions = 0
    gen = pl.scen_tree_gen
    orig_predict = rt.predict

    # live accounting inside the timed region: k_pair launch durations come from HIP events recorded on the
    # context stream around every launch (read back after the forward's own synchronisation point)
    def prof_predict(*a, **k):
        nonlocal pair_launch, pair_n2
        o = orig_predict(*a, **k)
        n, ms, pairs = rt.fusion_stats()
        a_off, l_off = a[1], a[3]
        n2 = sum(((a_off[i + 1] - a_off[i]) + (l_off[i + 1] - l_off[i]) + 1) ** 2 for i in range(len(a_off) - 1))
        nfl = sum(((a_off[i + 1] - a_off[i]) + (l_off[i + 1] - l_off[i]) + 1) * (a_off[i + 1] - a_off[i] + 1) for i in range(len(a_off) - 1))
        pair_ms.append(ms)
        pair_launch += n
        pair_n2 += n2
        # FLOPs the kernel actually issues on the MFMA (after the algebraic folds): per pair 2 x 128x128 GEMMs (65 536) +
        # 32 score MFMAs per 16 pairs (4 096); layer 4 updates only the actor/cls columns, layer 5 runs only those
        pair_exec[0] += 4 * n2 * 69632.0 + n2 * 36864.0 + nfl * 32768.0 + nfl * 36864.0
        return o

    rt.predict = prof_predict
    ctr0 = dict(pl.traj_tree_opt.counters)      # tree-iLQR accounting: the optimizer's own counters

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    n0 = pl.scen_tree_gen.n_expanded
    sim_steps = sim.run_plans(args.steps)            # K planning cycles = ~5K simulator steps
    expansions = pl.scen_tree_gen.n_expanded - n0
    barrier()
    dt = time.perf_counter() - t0
    rt.predict = orig_predict
    ctr = {k: v - ctr0.get(k, 0) for k, v in pl.traj_tree_opt.counters.items()}
    ilqr = {"trees": ctr["solves"], "iterations": ctr["iterations"]}
    rt.set_profiling(False)
    lcl = sim._observation()
    if dist is not None:
        red_dev = "cuda" if args.backend == "nccl" else "cpu"
        t = torch.tensor([dt], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        e = torch.tensor([expansions], device=red_dev, dtype=torch.float64)
        dist.all_reduce(e)
        expansions_all = float(e.item())
    else:
        expansions_all = float(expansions)
    total_pair_s = sum(pair_ms) * 1e-3
    achieved = F_MIN_N2 * pair_n2 / total_pair_s if total_pair_s > 0 else 0.0
    # HBM traffic of k_pair from the committed PMC passes of this same command (FETCH_SIZE, WRITE_SIZE in
    # separate rocprofv3 runs, gfx950 x2 correction on the fetch side): profiles/r01_pmc_k_pair.json
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_k_pair.json")
    if args.workload == "demo1" and os.path.exists(pmc_path):
        try:
            traffic = json.load(open(pmc_path))["k_pair_per_launch"]["hbm_bytes"]
        except Exception:
            traffic = None
    value = sim_steps * (1 if args.shard else world) / dt
    a = len(pl.agent_obs)
    l = gen.lane_feat_in.shape[0]
    out = {
        "metric": "sim steps/sec (whole node) + scenario-tree nodes expanded/sec, AV2 demo scenes",
        "value": value, "unit": "sim steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.shard else "weak", "vs_baseline": None,
        "dtype": "f32 predictor / f64 iLQR",
        "data": ("recorded AV2 scene (map + tracks of the reference's %s, tests/golden/scenes) with synthetic formula-initialised "
                 "weights" % args.workload) if real_scene else "synthetic",
        "nodes_expanded_per_s": expansions_all / dt,
        "config": {"workload": (f"recorded scene {args.workload}" if real_scene else f"{args.workload}-like synthetic scene") +
                               f": {a} agents x {l} lane polylines (N={a+l+1} tokens), "
                               f"one closed-loop planning cycle per step = AIME tree ({expansions // args.steps} node expansions, "
                               + ("the predictor's own modes with formula weights, exactly what the reference computes with these weights" if real_scene else
                                  "scripted mode branching on top of the real predictor forward: no trained checkpoint exists") + ") + "
                               f"tree-iLQR warm+full solves of {pl.timing['n_scen_trees']} scenario trees; closed loop: {sim_steps} simulator steps "
                               f"(0.02 s) for {args.steps} plans",
                   "agents": a, "lane_polylines": l, "expansions_per_plan": expansions // args.steps, "sim_steps_timed": sim_steps,
                   "scenario_trees_per_plan": pl.timing["n_scen_trees"], "parallelism": f"{world} independent scenes (one per GPU)"},
        "roofline": {"bound": "mfma", "achieved": achieved / 1e12, "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_F32_MFMA, "traffic": traffic, "traffic_unit": "bytes/launch (PMC, profiles/r01_pmc_k_pair.json)", "kernel": "k_pair (RelaFusionLayer pair kernel)",
                     "mfma_executed": {"tflops": (pair_exec[0] / total_pair_s / 1e12) if total_pair_s > 0 else None,
                                       "frac_of_peak": (pair_exec[0] / total_pair_s / PEAK_F32_MFMA) if total_pair_s > 0 else None,
                                       "note": "FLOPs actually issued on the MFMA after the algebraic folds (0.46-0.55 x F_min): frac "
                                               "above can exceed 1 because `achieved` prices the reference-minimal algorithm F_min, "
                                               "as SURVEY 8(d) prescribes"},
                     "launches_profiled": pair_launch, "avg_launch_ms": (sum(pair_ms) / pair_launch) if pair_launch else None,
                     "algorithmic_flops_per_launch": (F_MIN_N2 * pair_n2 / pair_launch) if pair_launch else None,
                     "note": "algorithmic FLOPs = SURVEY 8(d) F_min N^2 term (754944*N^2 per expansion over 6 launches); "
                             "launch durations from HIP events on the context stream, recorded inside the timed region"
                             + ("; measured in situ: launches that coincide with a speculative warm-start fit on the planner's second "
                                "context take ~4 us longer (frac 0.49 with MIND_SPECULATIVE_WARM_START=0, DESIGN 5)"
                                if pl.traj_tree_opt.speculative else "")},
        # the tree-iLQR kernel is latency-bound (serial depth x iterations): reported as rates, not against a roofline
        "ilqr": {"solves_per_s": ilqr["trees"] / dt, "iterations_per_s": ilqr["iterations"] / dt,
                 "trees_per_plan": ilqr["trees"] / max(args.steps, 1) / 2, "iterations_per_solve": ilqr["iterations"] / max(ilqr["trees"], 1),
                 "warm_start_fits_speculated": ctr["warm_speculated"], "warm_start_fits_reused": ctr["warm_hits"],
                 "note": "per rank; every scenario tree is solved twice per plan (warm start, then full cost); the warm-start fits of "
                         "the previous cycle's tree shapes run beside the predictor and are reused where the shape recurs"},
        "breakdown_ms": {"aime": pl.timing["aime_s"] * 1e3, "ilqr": pl.timing["ilqr_s"] * 1e3},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            scen_trees = gen.get_scenario_tree()
            out["cpu_baseline"] = cpu_baseline(pl, lcl, (a, l), expansions // args.steps, scen_trees)
        if world == 1 and not args.no_recorded and not real_scene:
            out["recorded_scenes"] = recorded_scenes()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
