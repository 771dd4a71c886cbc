#!/bin/bash
# tree-iLQR change: bit-exactness tests, then a same-box A/B of environment variants on the headline loop
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ilqr.py tests/test_gpu_ilqr_surface.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
for rep in 1 2; do
  i=0
  for v in "$@"; do
    env $v python bench.py --no-traffic --no-extras --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('k_ilqr', {}); print('variant $i [$v] rep $rep:', round(d['value'], 1), 'steps/s', round(d['ms_per_step'], 3), 'ms ilqr', round(d['breakdown_ms']['ilqr'], 3), 'kernel', round(k.get('kernel_ms_per_launch', 0), 3), k.get('phase_share'))
" | tee -a $O/ab.txt
    i=$((i+1))
  done
done
