#!/bin/bash
# Round-3 profile set: kernel stats (rocprofv3 --kernel-trace --stats) of the headline loop and of the full cfg4 tree, host-time split of both
O=gpurun_out/r03v; mkdir -p $O
export TMPDIR=/tmp
for wl in demo_1 cfg4tree; do
  steps=20; [ $wl = cfg4tree ] && steps=3
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline --no-extras --no-traffic > $GRAFT_REPO_ROOT/$O/bench_under_rocprof_$wl.json 2>/dev/null)
  f=$(find $O/trace_$wl -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$wl.csv; rm -rf $O/trace_$wl
  head -12 $O/kernel_stats_$wl.csv | cut -c1-70,140-230
done
timeout 300 python tools/gpu_time_host.py demo_1 40 formula_branching:20240121 > $O/host_time_demo_1.txt 2>&1; tail -32 $O/host_time_demo_1.txt
timeout 300 python tools/gpu_time_host.py cfg4tree 6 > $O/host_time_cfg4tree.txt 2>&1; tail -40 $O/host_time_cfg4tree.txt
timeout 600 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x 2>&1 | tail -2
