#!/bin/bash
O=gpurun_out/r02am; mkdir -p $O
for i in 1 2; do for os_ in 0 1; do
MIND_BENCH_OWN_STREAM=$os_ timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('demo_1 own_stream=$os_', round(d['value'],1), round(d['ms_per_step'],3), round(d['breakdown_ms']['aime'],3), round(d['breakdown_ms']['ilqr'],3))"
done; done
for os_ in 0 1; do
MIND_BENCH_OWN_STREAM=$os_ timeout 200 python bench.py --workload demo_1 --ckpt formula:20240121 --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('demo_1 plain own_stream=$os_', round(d['value'],1), round(d['ms_per_step'],3), round(d['breakdown_ms']['aime'],3), round(d['breakdown_ms']['ilqr'],3))"
MIND_BENCH_OWN_STREAM=$os_ timeout 300 python bench.py --workload cfg4tree --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4tree own_stream=$os_', round(d['ms_per_step'],2), round(d['nodes_expanded_per_s'],1), round(d['breakdown_ms']['aime'],2), round(d['breakdown_ms']['ilqr'],2))"
done
