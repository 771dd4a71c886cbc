#!/bin/bash
# GPU call r02x: merged host->device copies (tree-iLQR arena, re-basing, root inputs): suite + kernel trace (copy count) + bench
O=gpurun_out/r02x; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_demo1.json 2> $GRAFT_REPO_ROOT/$O/bench_demo1.err)
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_demo_1.csv; rm -rf $O/kt
grep -h "copyBuffer\|k_ilqr" $O/kernel_stats_demo_1.csv | cut -c1-140
python -c "import json; d=json.loads(open('$O/bench_demo1.json').read().strip().splitlines()[-1]); print('demo_1', d['value'], d['ms_per_step'], d['breakdown_ms'])"
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('demo_1 (no profiler)', d['value'], d['ms_per_step'], d['breakdown_ms'])"
