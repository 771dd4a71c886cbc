#!/bin/bash
# stage cycles of k_token (diagnostic build -DMIND_TOKEN_TRACE: block 0 of every launch prints its stage cycles) on the cfg4 full tree
O=gpurun_out/${1:-token_trace}; mkdir -p $O diag_build
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DMIND_TOKEN_TRACE mind_amd/csrc/mind_hip.hip -o diag_build/libmind_hip_token_trace.so
MIND_HIP_LIB=$GRAFT_REPO_ROOT/diag_build/libmind_hip_token_trace.so timeout 600 python bench.py --workload cfg4tree --steps 1 --warmup 1 --no-traffic --no-cpu-baseline --no-extras 2>&1 | grep "k_token mode" | tail -28 > $O/trace.txt
cat $O/trace.txt
