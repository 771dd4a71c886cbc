#!/bin/bash
# GPU call r03l: device-built root scene in the native AIME plan: equality tests, bench
O=gpurun_out/r03l; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_aime_native.py -m gpu -q -x 2>&1 | tail -25 > $O/pytest_native.txt; cat $O/pytest_native.txt
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['breakdown_ms']['aime'],d['breakdown_ms']['ilqr'],d['ilqr']['kernel_ms_per_launch'])"
