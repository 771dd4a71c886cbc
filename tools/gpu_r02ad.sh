#!/bin/bash
# GPU call r02ad: k_token at 128 VGPRs (two workgroups per CU): predictor tests, kernel trace on cfg4tree and demo_1
O=gpurun_out/r02ad; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_predictor.py tests/test_gpu_random_sweep.py -m gpu -q -x 2>&1 | tail -2
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kc -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4tree --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_cfg4tree_kt.json 2> $GRAFT_REPO_ROOT/$O/err.txt)
find $O/kc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_cfg4tree.csv; rm -rf $O/kc
grep -h "k_token" $O/kernel_stats_cfg4tree.csv | cut -c1-170
python -c "import json; d=json.loads(open('$O/bench_cfg4tree_kt.json').read().strip().splitlines()[-1]); print('cfg4tree', d['ms_per_step'], d['nodes_expanded_per_s'])"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_kt.json 2> $GRAFT_REPO_ROOT/$O/bench_kt.err)
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_demo_1.csv; rm -rf $O/kt
grep -h "k_token" $O/kernel_stats_demo_1.csv | cut -c1-170
