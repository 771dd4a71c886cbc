#!/bin/bash
# round 4, call f: the tile-native pair kernel in the library (bf16 edge tensor under plain bf16): parity + tree benches
O=gpurun_out/r04f; mkdir -p $O; export TMPDIR=/tmp
tools/micro/bin/pair_bench 24 321 7 2>&1 | grep -E "^pair_bench|^k_pair" | tee $O/pair_bench.txt
timeout 1200 python -m pytest tests/test_gpu_predictor.py tests/test_gpu_stress.py tests/test_gpu_random_sweep.py -m gpu -q -s 2>&1 | tail -12 | tee $O/pytest.txt
for wl in cfg4tree stress128tree; do for pv in bf16x3 bf16; do
  MIND_PAIR_PREC=$pv python bench.py --workload $wl --steps 8 --warmup 2 --no-traffic --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$wl $pv:', round(d['ms_per_step'], 2), 'ms per plan; aime', round(d['breakdown_ms']['aime'], 2), 'ilqr', round(d['breakdown_ms']['ilqr'], 2), '| pair hbm_frac', round(r['hbm']['frac'], 3), 'mfma_frac', round(r['mfma']['frac'], 3), 'avg launch ms', round(r['avg_launch_ms'], 3))
" | tee -a $O/tree_bench.txt
done; done
