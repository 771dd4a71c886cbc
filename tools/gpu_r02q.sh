#!/bin/bash
# GPU call r02q: pruning decisions on the device (k_aime_select): whole suite, host time per section, cfg4tree bench
O=gpurun_out/r02q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -6 $O/pytest.txt
timeout 200 python tools/gpu_time_host.py demo_1 40 formula_branching:20240121 > $O/host_time_demo_1.txt 2>&1
timeout 300 python tools/gpu_time_host.py cfg4tree 3 > $O/host_time_cfg4tree.txt 2>&1
grep -E "cycle|select|prune_select|aime_world|decide_branch|solve_batch|branch_aime" $O/host_time_demo_1.txt $O/host_time_cfg4tree.txt
timeout 300 python bench.py --workload cfg4tree --no-cpu-baseline --no-extras > $O/bench_cfg4tree.json 2> $O/bench_cfg4tree.err
python -c "import json; d=json.loads(open('$O/bench_cfg4tree.json').read().strip().splitlines()[-1]); print('cfg4tree', d['value'], d['ms_per_step'], d['nodes_expanded_per_s'])"
