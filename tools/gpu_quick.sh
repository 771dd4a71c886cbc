#!/bin/bash
# quick GPU check of a predictor-side change: predictor / AIME / plan parity tests + the headline bench line (no PMC passes)
O=gpurun_out/${1:-quick}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_predictor.py tests/test_gpu_aime_native.py tests/test_gpu_aime_golden.py tests/test_gpu_plan.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
timeout 600 python bench.py --no-traffic > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<PY
import json
for l in open("$O/bench.json"):
    if l.startswith("{"):
        d = json.loads(l); print(d["value"], d["ms_per_step"], d.get("tree", {}).get("ms_per_plan"), d.get("k_ilqr", {}).get("kernel_ms_per_launch"))
PY
