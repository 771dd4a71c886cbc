#!/bin/bash
O=gpurun_out/r02k; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
timeout 300 python bench.py --workload cfg4tree --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_cfg4tree.json 2> $O/cfg4.err
timeout 300 python tools/gpu_time_host.py cfg4tree 3 > $O/host_cfg4tree.txt 2>&1
timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 40 > $O/bench_demo_1.json 2> $O/bench.err
tail -3 $O/pytest.txt; grep -v amdgpu $O/host_cfg4tree.txt; python -c "
import json
for f in ('bench_demo_1','bench_cfg4tree'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['nodes_expanded_per_s'], d['breakdown_ms'])"
