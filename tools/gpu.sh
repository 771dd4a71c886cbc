#!/bin/bash
# The GPU-box runner: every measurement of profiles/ is one sub-command of this script (run it through gpurun from the repo root).
#   tools/gpu.sh <command> <tag> [args...]           results -> gpurun_out/<tag>/ (copy what is to be kept into profiles/)
#
#   final <tag>                         the whole -m gpu suite, smoke(), the default bench line, kernel stats + host-time split (demo_1, cfg4tree)
#   finalbench <tag>                    the same without the suite
#   quick <tag>                         predictor / AIME / plan parity tests + the headline bench line
#   tests <tag> <pytest args...>        pytest -m gpu on the given files / -k expression
#   kstats <tag> <workload> [ENV=..]    rocprofv3 --kernel-trace --stats over a short bench run of the workload (demo_1, cfg4tree, stress128tree ...)
#   pmc <tag> <kernel> <workload>       two --pmc passes (SQ activity | LDS + instruction mix), kernel trace only beside them, summarised per kernel
#   ab-env <tag> "<ENV=..>" ...         same-box A/B of environment variants on the headline loop (three interleaved repetitions)
#   ab-tree <tag> "<ENV=..>" ...        the same on the full cfg4 tree
#   micro <tag> pair|rmw [args]         tools/micro/bin/pair_bench (ablations of k_pair_t) / rmw_bench (read-modify-write streaming ceiling)
#   trace <tag> actor|dec|token         diagnostic build with -DMIND_{ACTOR,DEC,TOKEN}_TRACE: per-stage cycles printed by block 0
#   timeline <tag>                      GPU timeline (kernels + copies) of one planning cycle of the headline loop
#   config3 <tag>                       BASELINE config 3: demo_1..4 on one GPU as threads / fused rounds / pipelined / processes
#   ilqr-phase <tag> [workload] [plans] k_ilqr's own phase cycle counters per tree (and wave 0's fine-grained slots with a -DIL_PROFILE build)
#   shard-overhead <tag>                the sharded native plan on ONE rank: plain process vs a one-rank nccl group with every exchange executed
set -u
cmd=${1:?command}; tag=${2:?tag}; shift 2
O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
BENCH_LIGHT="--no-cpu-baseline --no-extras --no-traffic"

line() {  # one summary line of a bench JSON on stdin
  python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline') or {}
        print('$1', round(d['value'], 1), d['unit'], '|', round(d['ms_per_step'], 3), 'ms per step | aime', round(d['breakdown_ms']['aime'], 2), 'ilqr', round(d['breakdown_ms']['ilqr'], 2),
              '| pair hbm_frac', round(r.get('hbm_frac', 0), 3), 'avg launch ms', round(r.get('avg_launch_ms', 0), 3), '| k_ilqr ms', round((d.get('k_ilqr') or {}).get('kernel_ms_per_launch', 0), 3))
"
}

case $cmd in
final|finalbench)
  if [ $cmd = final ]; then
  ( time timeout 2400 python -m pytest tests -m gpu -q -s ) > $O/pytest_gpu_full.txt 2>&1; grep -n "passed\|failed\|^real" $O/pytest_gpu_full.txt | tail -3
  grep "^\[demo\|^\.\[demo\|same tree chosen\|cycles agree outright\|^demo_\|^\.demo_\|^stress" $O/pytest_gpu_full.txt | sed 's/^\.*//' > $O/pytest_gpu_parity_lines.txt
  tail -25 $O/pytest_gpu_full.txt > $O/pytest_gpu.txt
  fi
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  ( time MIND_BENCH_EXTRAS=$ROOT/$O/bench_extras.json timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
  tail -2 $O/bench.err; wc -c $O/bench.json
  python - <<PY
import json
d = json.load(open("$O/bench_extras.json"))
print(d["value"], d["ms_per_step"], d["breakdown_ms"]["aime"], d["breakdown_ms"]["ilqr"])
print("roofline", {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "frac", "traffic")})
print("k_ilqr", d["k_ilqr"]["kernel_ms_per_launch"], d["k_ilqr"]["phase_share"])
print("dtype", d["dtype"][:40], "| bf16x3", (d.get("bf16x3") or {}).get("value"))
for k in ("tree_f32", "tree", "stress", "stress_bf16", "stress_deep", "stress_deeper", "synthetic_branching", "plain_formula_weights"):
    t = d.get(k) or {}
    print(k, t.get("ms_per_plan"), t.get("nodes_expanded_per_s"), t.get("aime_native_plans"), (t.get("k_pair") or {}).get("hbm_frac"), t.get("error"))
print("recorded", {k: v.get("sim_steps_per_s") for k, v in (d.get("recorded_scenes") or {}).items() if isinstance(v, dict)})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
  for wl in demo_1 cfg4tree; do
    steps=20; [ $wl = cfg4tree ] && steps=3
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/trace_$wl -- python $ROOT/bench.py --workload $wl --steps $steps --warmup 2 $BENCH_LIGHT > $ROOT/$O/bench_under_rocprof_$wl.json 2>/dev/null)
    f=$(find $O/trace_$wl -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$wl.csv; rm -rf $O/trace_$wl
    head -10 $O/kernel_stats_$wl.csv | cut -c1-60,140-230
  done
  timeout 300 python tools/gpu_time_host.py demo_1 40 formula_branching:20240121 > $O/host_time_demo_1.txt 2>&1; tail -30 $O/host_time_demo_1.txt
  timeout 300 python tools/gpu_time_host.py cfg4tree 6 > $O/host_time_cfg4tree.txt 2>&1; tail -30 $O/host_time_cfg4tree.txt
  ;;
quick)
  timeout 900 python -m pytest tests/test_gpu_predictor.py tests/test_gpu_aime_native.py tests/test_gpu_aime_golden.py tests/test_gpu_plan.py -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest.txt
  timeout 600 python bench.py --no-traffic > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; line headline < $O/bench.json
  ;;
tests)
  timeout 2400 python -m pytest "$@" -m gpu -q -s 2>&1 | tail -40 | tee $O/pytest.txt
  ;;
kstats)
  wl=${1:-demo_1}; shift || true
  steps=20; case $wl in cfg4tree|stress128tree) steps=3;; stressdeep) steps=1;; esac
  (cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/trace -- python $ROOT/bench.py --workload $wl --steps $steps --warmup 2 $BENCH_LIGHT > $ROOT/$O/bench.json 2> $ROOT/$O/bench.err)
  f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$wl.csv; rm -rf $O/trace
  python - <<PY
import csv
for r in list(csv.DictReader(open("$O/kernel_stats_$wl.csv")))[:16]:
    print(r["Name"][:72].ljust(72), r["Calls"].rjust(6), ("%.1f us avg" % (float(r["AverageNs"]) / 1e3)).rjust(14), (r["Percentage"] + " %").rjust(10), ("max %.1f us" % (float(r["MaxNs"]) / 1e3)).rjust(16))
PY
  ;;
pmc)
  kern=${1:?kernel name}; wl=${2:-demo_1}
  steps=20; case $wl in cfg4tree|stress128tree) steps=2;; esac
  (cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $ROOT/$O/sq -- python $ROOT/bench.py --workload $wl --steps $steps --warmup 1 $BENCH_LIGHT > /dev/null 2>&1)
  (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $ROOT/$O/lds -- python $ROOT/bench.py --workload $wl --steps $steps --warmup 1 $BENCH_LIGHT > /dev/null 2>&1)
  python tools/pmc_summary.py $O/sq $kern > $O/pmc_sq_$kern.json 2>&1
  python tools/pmc_summary.py $O/lds $kern > $O/pmc_lds_$kern.json 2>&1
  rm -rf $O/sq $O/lds
  python - <<PY
import json
for f in ("$O/pmc_sq_$kern.json", "$O/pmc_lds_$kern.json"):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable:", e); print(open(f).read()[:500]); continue
    for k, v in d.items():
        print(k, "avg_us", round(v.get("avg_us", 0), 1), {c: round(x["per_dispatch"]) for c, x in v["counters"].items()}, v.get("derived"))
PY
  ;;
ab-env|ab-tree)
  if [ $cmd = ab-env ]; then W="--steps 40 --warmup 5"; R="1 2 3"; python bench.py $BENCH_LIGHT --steps 5 --warmup 1 > /dev/null 2>&1; else W="--workload cfg4tree --steps 8 --warmup 2"; R="1 2"; fi
  for rep in $R; do i=0
    for v in "$@"; do env $v python bench.py $BENCH_LIGHT $W 2>/dev/null | line "variant $i [$v] rep $rep:" | tee -a $O/ab.txt; i=$((i+1)); done
  done
  ;;
micro)
  which=${1:?pair|rmw}; shift
  timeout 120 tools/micro/bin/${which}_bench "$@" > $O/${which}_bench.txt 2>&1; grep -v "^\[k_pair" $O/${which}_bench.txt
  ;;
trace)
  what=${1:?actor|dec|token}
  D=$(echo $what | tr a-z A-Z); mkdir -p diag_build
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DMIND_${D}_TRACE mind_amd/csrc/mind_hip.hip -o diag_build/libmind_hip_${what}_trace.so
  MIND_HIP_LIB=$ROOT/diag_build/libmind_hip_${what}_trace.so timeout 600 python -m pytest tests/test_gpu_predictor.py -m gpu -q -x -s -k "golden" 2>&1 | grep "^\[k_" | head -12 | tee $O/trace.txt
  ;;
timeline)
  (cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $ROOT/$O/trace -- python $ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras > $ROOT/$O/bench.json 2> $ROOT/$O/bench.err)
  python tools/gpu_timeline.py $O/trace > $O/timeline.txt 2>&1; tail -90 $O/timeline.txt
  find $O/trace -name "*.csv" -size +2M -delete
  ;;
config3)
  for mode in "" "--fused" "--pipelined" "--processes"; do
    timeout 400 python bench.py --workload demo_all --concurrent 4 $mode --steps 40 --warmup 5 $BENCH_LIGHT 2>$O/err.txt | tail -1 > $O/line.json
    python -c "import json; d=json.loads(open('$O/line.json').read()); print('demo_all x4 [$mode]', round(d['value'],1), 'sim steps/s', round(d['ms_per_step'],3), 'ms per round of plans')" || tail -3 $O/err.txt
    cat $O/line.json >> $O/config3.jsonl
  done
  ;;
shard-overhead)
  for mode in plain forced; do
    if [ $mode = forced ]; then X="MIND_FORCE_COLLECTIVES=1 MIND_DIST_BACKEND=nccl"; else X=""; fi
    env $X RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29591 MIND_TEST_WORKLOAD=cfg4tree python tests/dist_gpu_worker.py $O/$mode.pkl 8 > $O/$mode.log 2>&1
    python - <<PY
import pickle
d = pickle.load(open("$O/$mode.pkl", "rb"))
w = d["wall_ms"][2:]
print("$mode: ms per plan", w, "mean", round(sum(w) / len(w), 2), "| collectives", d["collectives"], "gathered MB", round(d["gathered"] / 1e6, 1), "| aime ms/plan", round(d["timing"]["aime_s"] / d["timing"]["plans"] * 1e3, 2))
PY
  done | tee $O/summary.txt
  ;;
ilqr-phase)
  # per-phase cycle counters of k_ilqr (stats of every tree of a plan) on the headline loop; with a -DIL_PROFILE build (diag_build/libmind_hip_ilprof.so,
  # compiled in the build container: hipcc ... -DIL_PROFILE) also the fine-grained slots of wave 0
  wl=${1:-demo_1}; n=${2:-2}
  timeout 300 python tools/gpu_ilqr_phase.py $wl $n formula_branching:20240121 2>&1 | grep "^\[k_ilqr" > $O/ilqr_phase.txt; tail -12 $O/ilqr_phase.txt
  if [ -f diag_build/libmind_hip_ilprof.so ]; then
    MIND_HIP_LIB=$ROOT/diag_build/libmind_hip_ilprof.so timeout 300 python tools/gpu_ilqr_phase.py $wl $n formula_branching:20240121 2>&1 | grep "^\[k_ilqr" > $O/ilqr_phase_prof.txt; tail -24 $O/ilqr_phase_prof.txt
  fi
  ;;
*) echo "unknown command $cmd"; exit 2;;
esac
