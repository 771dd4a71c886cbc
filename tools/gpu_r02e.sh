#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O
export TMPDIR=/tmp
for v in trace trace_top; do
  echo "== $v" >> $O/trace.txt
  MIND_HIP_LIB=$PWD/mind_amd/libmind_hip_$v.so timeout 120 python tests/diag/gpu_diag_predictor.py --prec bf16x3 --timing-only 2>&1 | grep -E "k_pair_bf<1,3> um=0|timing" | tail -8 >> $O/trace.txt
done
cat $O/trace.txt
