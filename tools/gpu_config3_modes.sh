#!/bin/bash
# BASELINE config 3 (demo_1..4 concurrently on one GPU): threads (one context per scene), fused rounds, processes
O=gpurun_out/${1:-config3}; mkdir -p $O
export TMPDIR=/tmp
for mode in "" "--fused" "--pipelined" "--processes"; do
timeout 400 python bench.py --workload demo_all --concurrent 4 $mode --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-traffic 2>$O/err.txt | tail -1 > $O/line.json
python -c "import json; d=json.loads(open('$O/line.json').read()); print('demo_all x4 [$mode]', round(d['value'],1), 'sim steps/s', round(d['ms_per_step'],3), 'ms per round of plans')" || tail -3 $O/err.txt
cat $O/line.json >> $O/config3.jsonl
done
