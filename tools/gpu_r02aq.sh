#!/bin/bash
# GPU call r02aq: decoder halves on two streams: tests, same-box A/B of the headline loop, kernel trace
O=gpurun_out/r02aq; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_predictor.py -m gpu -q -x 2>&1 | tail -1
for i in 1 2 3; do for ov in 1 0; do
MIND_DEC_OVERLAP=$ov timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('demo_1 dec_overlap=$ov', round(d['value'],1), round(d['ms_per_step'],3), round(d['breakdown_ms']['aime'],3))"
done; done
