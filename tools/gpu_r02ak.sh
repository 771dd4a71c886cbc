#!/bin/bash
O=gpurun_out/r02ak; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_plan.py tests/test_gpu_aime_world.py -m gpu -q -x -k "windows" 2>&1 | tail -1
for sp in 1 0; do for m in 1 0; do
SPEC=$sp MIND_DEVICE_WINDOWS=$m timeout 300 python tools/gpu_time_host.py cfg4tree 4 > $O/ht_$sp$m.txt 2>&1; echo "speculative=$sp device_windows=$m"; grep -E "cycle|C: mind_aime_rebase|update_obser_batch|branch_aime" $O/ht_$sp$m.txt | tr '\n' ' '; echo
done; done
