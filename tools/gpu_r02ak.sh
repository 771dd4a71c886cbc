#!/bin/bash
O=gpurun_out/r02ak; mkdir -p $O
for sp in 1 0; do for m in 1 0; do
SPEC=$sp MIND_DEVICE_WINDOWS=$m timeout 300 python tools/gpu_time_host.py cfg4tree 4 > $O/ht_$sp$m.txt 2>&1; echo "speculative=$sp device_windows=$m"; grep -E "cycle|C: mind_aime_rebase|update_obser_batch|branch_aime|solve_batch" $O/ht_$sp$m.txt | tr '\n' ' '; echo
done; done
