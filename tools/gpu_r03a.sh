#!/bin/bash
# GPU call r03a: baseline of the round-2 tree: per-phase cycles of k_ilqr (IL_PROFILE build), host-time split, default bench
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
MIND_HIP_LIB=build/libmind_hip_prof.so timeout 300 python tools/gpu_ilqr_phase.py demo_1 2 formula_branching:20240121 > $O/ilqr_phase.txt 2>&1
timeout 300 python tools/gpu_time_host.py demo_1 30 formula_branching:20240121 > $O/host_time.txt 2>&1
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
tail -5 $O/ilqr_phase.txt; cat $O/host_time.txt | head -40; python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['breakdown_ms'],d['ilqr'])"
