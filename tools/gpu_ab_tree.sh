#!/bin/bash
# same-box A/B of environment variants on the cfg4 full-tree plan: tools/gpu_ab_tree.sh <tag> "<ENV=..>" "<ENV=..>" ...
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  i=0
  for v in "$@"; do
    env $v python bench.py --workload cfg4tree --steps 8 --warmup 2 --no-traffic --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('variant $i [$v] rep $rep:', round(d['ms_per_step'], 2), 'ms per plan; aime', round(d['breakdown_ms']['aime'], 2), 'ilqr', round(d['breakdown_ms']['ilqr'], 2))
" | tee -a $O/ab.txt
    i=$((i+1))
  done
done
