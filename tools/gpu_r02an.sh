#!/bin/bash
O=gpurun_out/r02an; mkdir -p $O
for i in 1 2; do
timeout 600 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err; echo "rc=$?"
python -c "import json; d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); print('demo_1', round(d['value'],1), round(d['ms_per_step'],3), 'tree', round(d['tree']['ms_per_plan'],2), round(d['tree']['nodes_expanded_per_s'],1), 'cpu', round(d['cpu_baseline']['value'],2))"
done
