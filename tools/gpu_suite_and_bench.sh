#!/bin/bash
# GPU call r03k: stress workload test, full bench line with the new blocks (k_ilqr, exact_fp32, traffic, stress)
O=gpurun_out/r03k; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stress.py tests/test_gpu_plan.py -m gpu -q -x -s 2>&1 | tail -12 > $O/pytest.txt; cat $O/pytest.txt
( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -4
tail -3 $O/bench.err
python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'],d['ms_per_step'],d['breakdown_ms']['aime'],d['breakdown_ms']['ilqr'])
print('roofline', {k:d['roofline'][k] for k in ('bound','achieved','peak','frac','traffic')}, d['roofline'].get('traffic_detail'))
print('k_ilqr', d['k_ilqr'])
print('exact_fp32', d.get('exact_fp32'))
print('tree', {k:d['tree'].get(k) for k in ('ms_per_plan','nodes_expanded_per_s','plans_timed')}, d['tree'].get('k_pair'))
print('stress', {k:d['stress'].get(k) for k in ('ms_per_plan','nodes_expanded_per_s','error')}, (d['stress'].get('k_ilqr') or {}).get('sweep_gb_per_s'))
print('stress_bf16', {k:d.get('stress_bf16',{}).get(k) for k in ('ms_per_plan','nodes_expanded_per_s')})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
