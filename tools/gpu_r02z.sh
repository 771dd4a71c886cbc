#!/bin/bash
O=gpurun_out/r02z; mkdir -p $O
export TMPDIR=/tmp
export MIND_HIP_LIB=$GRAFT_REPO_ROOT/mind_amd/libmind_hip_xcd.so
timeout 300 python -m pytest tests/test_gpu_ilqr.py -m gpu -q -x 2>&1 | tail -2
for G in 8 16; do
(cd /tmp && MIND_ILQR_WGS=$G timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kc -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4tree --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_cfg4tree_xcd_g$G.json 2> $GRAFT_REPO_ROOT/$O/err.txt)
find $O/kc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_cfg4tree_xcd_g$G.csv; rm -rf $O/kc
echo "same-XCD release, G=$G"; grep -h "k_ilqr<false, true" $O/kernel_stats_cfg4tree_xcd_g$G.csv | cut -c1-150
python -c "import json; d=json.loads(open('$O/bench_cfg4tree_xcd_g$G.json').read().strip().splitlines()[-1]); print('cfg4tree', d['ms_per_step'], d['nodes_expanded_per_s'], d['breakdown_ms']['aime'], d['breakdown_ms']['ilqr'])"
done
