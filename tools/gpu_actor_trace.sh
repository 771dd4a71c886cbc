#!/bin/bash
# per-stage cycle trace of k_actor_mfma: diagnostic build (hipcc ... -DMIND_ACTOR_TRACE -o diag_build/libmind_hip_actor_trace.so), block 0 prints
# the cycles of every conv / GroupNorm stage
O=gpurun_out/${1:-actor_trace}; mkdir -p $O
export TMPDIR=/tmp
mkdir -p diag_build && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DMIND_ACTOR_TRACE mind_amd/csrc/mind_hip.hip -o diag_build/libmind_hip_actor_trace.so
MIND_HIP_LIB=$GRAFT_REPO_ROOT/diag_build/libmind_hip_actor_trace.so timeout 600 python -m pytest tests/test_gpu_predictor.py -m gpu -q -x -s -k "golden" 2>&1 | grep "k_actor_mfma" | head -8 > $O/trace.txt
cat $O/trace.txt
