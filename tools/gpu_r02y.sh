#!/bin/bash
# GPU call r02y: multi-workgroup k_ilqr for wide cost trees: bit-exactness tests, cfg4tree kernel trace + bench
O=gpurun_out/r02y; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ilqr.py tests/test_gpu_ilqr_surface.py tests/test_gpu_random_sweep.py -m gpu -q -x > $O/pytest_ilqr.txt 2>&1; echo "rc=$?" >> $O/pytest_ilqr.txt; tail -5 $O/pytest_ilqr.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kc -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4tree --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_cfg4tree_kt.json 2> $GRAFT_REPO_ROOT/$O/bench_cfg4tree_kt.err)
find $O/kc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_cfg4tree.csv; rm -rf $O/kc
grep -h "k_ilqr" $O/kernel_stats_cfg4tree.csv | cut -c1-170
timeout 300 python bench.py --workload cfg4tree --no-cpu-baseline --no-extras > $O/bench_cfg4tree.json 2> $O/bench_cfg4tree.err
python -c "import json; d=json.loads(open('$O/bench_cfg4tree.json').read().strip().splitlines()[-1]); print('cfg4tree', d['value'], d['ms_per_step'], d['nodes_expanded_per_s'], d['breakdown_ms'])"
