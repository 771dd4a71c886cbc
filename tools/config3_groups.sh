#!/bin/bash
# BASELINE config 3 as groups: P scenes = P / Q host processes of Q scenes each (bench.py --concurrent P --processes --per-process Q)
for cfg in "$@"; do
  set -- $cfg; P=$1; Q=$2; S=${3:-10}
  MIND_ILQR_SLOTS=$S timeout 600 python bench.py --workload demo_all --concurrent $P --processes --per-process $Q --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-traffic 2>/tmp/err.txt | tail -1 > /tmp/line.json
  python -c "import json; d=json.loads(open('/tmp/line.json').read()); print('demo_all x$P in groups of $Q (slots $S):', round(d['value'],1), 'sim steps/s', round(d['ms_per_step'],3), 'ms per round of plans')" || tail -5 /tmp/err.txt
done
