#!/bin/bash
# GPU call r02p: pre-split ActorNet MFMA kernel + host-side changes: predictor/plan tests, kernel-trace A/B, bench
O=gpurun_out/r02p; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -6 $O/pytest.txt
for m in 6 3; do
  export MIND_ACTOR_SPLIT=$m
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt$m -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_demo1_split$m.json 2> $GRAFT_REPO_ROOT/$O/bench_demo1_split$m.err)
  find $O/kt$m -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_demo_1_split$m.csv; rm -rf $O/kt$m
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kc$m -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4tree --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_cfg4tree_split$m.json 2> $GRAFT_REPO_ROOT/$O/bench_cfg4tree_split$m.err)
  find $O/kc$m -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_cfg4tree_split$m.csv; rm -rf $O/kc$m
  grep -h "k_actor" $O/kernel_stats_demo_1_split$m.csv $O/kernel_stats_cfg4tree_split$m.csv | cut -c1-160
done
unset MIND_ACTOR_SPLIT
timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_demo_1.json 2>$O/bench_demo_1.err
python -c "import json; d=json.loads(open('$O/bench_demo_1.json').read().strip().splitlines()[-1]); print('demo_1', d['value'], d['ms_per_step'], d['breakdown_ms'])"
timeout 200 python tools/gpu_time_host.py demo_1 40 formula_branching:20240121 > $O/host_time_demo_1.txt 2>&1
