#!/bin/bash
# per-stage cycle trace of k_dec_scene (diagnostic build -DMIND_DEC_TRACE): block 0 prints the cycles between its dense / LayerNorm stages
O=gpurun_out/${1:-dec_trace}; mkdir -p $O diag_build
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DMIND_DEC_TRACE mind_amd/csrc/mind_hip.hip -o diag_build/libmind_hip_dec_trace.so
MIND_HIP_LIB=$GRAFT_REPO_ROOT/diag_build/libmind_hip_dec_trace.so timeout 600 python -m pytest tests/test_gpu_predictor.py -m gpu -q -x -s -k "golden" 2>&1 | grep "k_dec_scene" | head -6 > $O/trace.txt
cat $O/trace.txt
