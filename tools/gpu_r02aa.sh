#!/bin/bash
# GPU call r02aa: multi-workgroup k_ilqr as the default for wide trees: whole suite, benches
O=gpurun_out/r02aa; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
timeout 300 python bench.py --workload cfg4tree --no-cpu-baseline --no-extras > $O/bench_cfg4tree.json 2> $O/bench_cfg4tree.err
python -c "import json; d=json.loads(open('$O/bench_cfg4tree.json').read().strip().splitlines()[-1]); print('cfg4tree', d['value'], d['ms_per_step'], d['nodes_expanded_per_s'], d['breakdown_ms'], d['ilqr'])"
timeout 300 python bench.py --workload stress128tree --no-cpu-baseline --no-extras > $O/bench_stress128tree.json 2> $O/bench_stress.err
python -c "import json; d=json.loads(open('$O/bench_stress128tree.json').read().strip().splitlines()[-1]); print('stress128tree', d['value'], d['ms_per_step'], d['nodes_expanded_per_s'], d['breakdown_ms'])"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kc -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4tree --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_cfg4tree_kt.json 2> $GRAFT_REPO_ROOT/$O/err.txt)
find $O/kc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_cfg4tree.csv; rm -rf $O/kc
head -8 $O/kernel_stats_cfg4tree.csv | cut -c1-150
