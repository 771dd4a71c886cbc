#!/bin/bash
# GPU call r03p: token kernel on the full cfg4 tree: VALU vs fp32 MFMA (plan time and average launch)
O=gpurun_out/r03p; mkdir -p $O
export TMPDIR=/tmp
for tm in 0 1; do
MIND_TOK_MFMA=$tm timeout 300 python bench.py --workload cfg4tree --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4tree tok_mfma=$tm', round(d['value'],1), round(d['ms_per_step'],3), round(d['breakdown_ms']['aime'],3))"
(cd /tmp && MIND_TOK_MFMA=$tm rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace$tm -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4tree --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-traffic > /dev/null 2>&1)
f=$(find $O/trace$tm -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_cfg4tree_tokmfma$tm.csv; grep -i "token\|Name" $O/kernel_stats_cfg4tree_tokmfma$tm.csv | cut -c1-60,150-260; rm -rf $O/trace$tm
done
