#!/bin/bash
# GPU call r03o: fp32-MFMA token kernel: predictor parity, A/B of the headline loop (tok_mfma 1 / 0), kernel stats
O=gpurun_out/r03o; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_predictor.py tests/test_gpu_random_sweep.py -m gpu -q -x 2>&1 | tail -12 > $O/pytest_pred.txt; cat $O/pytest_pred.txt
for tm in 1 0 1 0; do
MIND_TOK_MFMA=$tm timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('demo_1 tok_mfma=$tm', round(d['value'],1), round(d['ms_per_step'],3), round(d['breakdown_ms']['aime'],3))"
done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-traffic > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_demo_1.csv; head -14 $O/kernel_stats_demo_1.csv | cut -c1-150; rm -rf $O/trace
