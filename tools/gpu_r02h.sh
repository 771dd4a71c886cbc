#!/bin/bash
O=gpurun_out/r02h; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
timeout 300 python tools/gpu_time_host.py cfg4tree 3 > $O/host_cfg4tree.txt 2>&1
timeout 120 python tools/gpu_time_host.py demo_1 20 formula_branching:20240121 > $O/host_demo_1_branching.txt 2>&1
timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 40 > $O/bench_demo_1.json 2> $O/bench.err
tail -3 $O/pytest.txt; grep -v amdgpu $O/host_cfg4tree.txt; grep -v amdgpu $O/host_demo_1_branching.txt; python -c "
import json; d=json.load(open('$O/bench_demo_1.json')); print('demo_1 value', d['value'], d['ms_per_step'], d['breakdown_ms'])"
