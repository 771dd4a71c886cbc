#!/bin/bash
# GPU call r02aj: re-basing windows assembled on the device: new tests, suite, host time on cfg4tree, benches
O=gpurun_out/r02aj; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_aime_world.py tests/test_gpu_plan.py -m gpu -q -x -k "windows or rebase" > $O/pytest_new.txt 2>&1; tail -5 $O/pytest_new.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
timeout 300 python tools/gpu_time_host.py cfg4tree 3 > $O/host_time_cfg4tree.txt 2>&1; grep -E "cycle|aime_rebase|update_obser_batch|decide_branch|branch_aime" $O/host_time_cfg4tree.txt
timeout 300 python bench.py --workload cfg4tree --no-cpu-baseline --no-extras > $O/bench_cfg4tree.json 2> $O/err.txt
python -c "import json; d=json.loads(open('$O/bench_cfg4tree.json').read().strip().splitlines()[-1]); print('cfg4tree', d['ms_per_step'], d['nodes_expanded_per_s'], d['breakdown_ms']['aime'], d['breakdown_ms']['ilqr'])"
