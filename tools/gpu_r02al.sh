#!/bin/bash
O=gpurun_out/r02al; mkdir -p $O
for i in 1 2; do for sp in 1 0; do
MIND_SPECULATIVE_WARM_START=$sp timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('demo_1 speculative=$sp', round(d['value'],1), round(d['ms_per_step'],3), d['breakdown_ms']['aime'], d['breakdown_ms']['ilqr'], d['ilqr']['warm_start_fits_speculated'], d['ilqr']['warm_start_fits_reused'])"
done; done
for sp in 1 0; do
MIND_SPECULATIVE_WARM_START=$sp timeout 200 python bench.py --workload demo_1 --ckpt formula:20240121 --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('demo_1 plain weights speculative=$sp', round(d['value'],1), round(d['ms_per_step'],3), d['breakdown_ms']['aime'], d['breakdown_ms']['ilqr'], d['ilqr']['warm_start_fits_speculated'], d['ilqr']['warm_start_fits_reused'])"
done
