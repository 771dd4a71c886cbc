#!/bin/bash
# A/B of environment toggles on ONE box: tools/gpu_ab_env.sh <tag> "<ENV=.. ENV=..>" "<ENV=..>" ...   (each variant twice, interleaved)
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
python bench.py --no-traffic --no-extras --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1    # page the image in
for rep in 1 2 3; do
  i=0
  for v in "$@"; do
    env $v python bench.py --no-traffic --no-extras --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('variant $i [$v] rep $rep:', round(d['value'], 1), 'steps/s', round(d['ms_per_step'], 3), 'ms', d['breakdown_ms']['aime'], d['breakdown_ms']['ilqr'])
" | tee -a $O/ab.txt
    i=$((i+1))
  done
done
