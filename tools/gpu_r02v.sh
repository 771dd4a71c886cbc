#!/bin/bash
O=gpurun_out/r02v; mkdir -p $O
export TMPDIR=/tmp
for i in noinline; do
timeout 300 python -m pytest tests/test_gpu_ilqr.py tests/test_gpu_ilqr_surface.py -m gpu -q -x 2>&1 | tail -1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_demo1_$i.json 2> $GRAFT_REPO_ROOT/$O/bench_demo1.err)
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_demo_1_$i.csv; rm -rf $O/kt
grep -h "k_ilqr" $O/kernel_stats_demo_1_$i.csv | cut -c1-140
python -c "import json; d=json.loads(open('$O/bench_demo1_$i.json').read().strip().splitlines()[-1]); print('demo_1', d['value'], d['ms_per_step'], d['breakdown_ms'])"
done
