#!/bin/bash
O=gpurun_out/r02ag; mkdir -p $O
timeout 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?"
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('demo_1', d['value'], d['ms_per_step'], d['ilqr']['kernel_ms_per_launch'], d['ilqr']['kernel_launches_timed'], d['ilqr']['workgroups_per_tree']); print('tree', d['tree']['ms_per_plan'], d['tree']['nodes_expanded_per_s'], d['tree']['k_ilqr_ms_per_launch'], d['tree']['k_ilqr_workgroups_per_tree'])"
timeout 300 python -m pytest tests/test_gpu_plan.py -m gpu -q -k "bench" 2>&1 | tail -2
