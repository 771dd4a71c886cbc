#!/bin/bash
# native AIME plan on the scripted workloads: parity test + the cfg4 full-tree bench line
O=gpurun_out/${1:-native_tree}; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_aime_native.py tests/test_gpu_stress.py tests/test_gpu_sharded.py -k "not nothing" -m gpu -q -x 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
timeout 600 python bench.py --workload cfg4tree --steps 8 --warmup 2 --no-traffic --no-cpu-baseline --no-extras > $O/bench_cfg4tree.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<PY
import json
for l in open("$O/bench_cfg4tree.json"):
    if l.startswith("{"):
        d = json.loads(l); print(d["value"], d["unit"], d["ms_per_step"], d.get("breakdown_ms"), d.get("native_plans"))
PY
