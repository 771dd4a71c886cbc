#!/bin/bash
# GPU call r03e: rollout two nodes per trip + deferred stores: bit-exactness, coarse phase cycles, bench
O=gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ilqr.py tests/test_gpu_ilqr_surface.py tests/test_gpu_random_sweep.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest_ilqr.txt; cat $O/pytest_ilqr.txt
timeout 300 python tools/gpu_ilqr_phase.py demo_1 2 formula_branching:20240121 > $O/ilqr_phase.txt 2>&1
grep "k_ilqr" $O/ilqr_phase.txt | tail -10
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['breakdown_ms']['aime'],d['breakdown_ms']['ilqr'],d['ilqr']['kernel_ms_per_launch'])"
