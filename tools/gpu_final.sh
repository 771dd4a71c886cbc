#!/bin/bash
# end-of-round GPU call: the whole -m gpu suite, smoke(), the default bench line (all blocks), kernel stats + host-time split of the headline
# loop and of the full cfg4 tree.  tools/gpu_final.sh <tag>
O=gpurun_out/${1:-final}; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x -s ) > $O/pytest_gpu_full.txt 2>&1; grep -n "passed\|failed\|^real" $O/pytest_gpu_full.txt | tail -3
grep "^\[demo\|^\.\[demo\|same tree chosen\|cycles agree outright" $O/pytest_gpu_full.txt | sed 's/^\.//' > $O/pytest_gpu_parity_lines.txt
tail -25 $O/pytest_gpu_full.txt > $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
tail -2 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["breakdown_ms"]["aime"], d["breakdown_ms"]["ilqr"])
print("roofline", {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "frac", "traffic")})
print("k_ilqr", d["k_ilqr"]["kernel_ms_per_launch"], d["k_ilqr"]["phase_share"])
print("exact_fp32", (d.get("exact_fp32") or {}).get("value"))
for k in ("tree", "stress", "stress_bf16", "synthetic_branching", "plain_formula_weights"):
    t = d.get(k) or {}
    print(k, t.get("ms_per_plan"), t.get("nodes_expanded_per_s"), t.get("aime_native_plans"), (t.get("k_pair") or {}).get("hbm_frac"), t.get("error"))
print("recorded", {k: v.get("sim_steps_per_s") for k, v in (d.get("recorded_scenes") or {}).items() if isinstance(v, dict)})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
for wl in demo_1 cfg4tree; do
  steps=20; [ $wl = cfg4tree ] && steps=3
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline --no-extras --no-traffic > $GRAFT_REPO_ROOT/$O/bench_under_rocprof_$wl.json 2>/dev/null)
  f=$(find $O/trace_$wl -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$wl.csv; rm -rf $O/trace_$wl
  head -10 $O/kernel_stats_$wl.csv | cut -c1-60,140-230
done
timeout 300 python tools/gpu_time_host.py demo_1 40 formula_branching:20240121 > $O/host_time_demo_1.txt 2>&1; tail -30 $O/host_time_demo_1.txt
timeout 300 python tools/gpu_time_host.py cfg4tree 6 > $O/host_time_cfg4tree.txt 2>&1; tail -30 $O/host_time_cfg4tree.txt
