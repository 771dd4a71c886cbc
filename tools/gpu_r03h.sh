#!/bin/bash
# GPU call r03h: whole GPU suite with the native AIME plan + k_ilqr changes; default bench line
O=gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['breakdown_ms'],d['roofline'] and d['roofline']['frac'], d.get('tree',{}).get('ms_per_plan'))"
