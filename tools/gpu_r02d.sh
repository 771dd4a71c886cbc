#!/bin/bash
# GPU call r02d: ping-pong (barrier-locked wave pairs) variants of the bf16 pair kernel
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
for v in base pp pp_top; do
  echo "== variant $v" >> $O/ab.txt
  MIND_HIP_LIB=$PWD/mind_amd/libmind_hip_$v.so timeout 90 python tests/diag/gpu_diag_predictor.py --prec bf16x3,bf16 --timing-only --big 2>&1 | grep -E "timing|arith" >> $O/ab.txt
  echo "rc=$?" >> $O/ab.txt
done
echo "== parity of variant pp" >> $O/parity.txt
MIND_HIP_LIB=$PWD/mind_amd/libmind_hip_pp.so timeout 300 python -m pytest tests/test_gpu_predictor.py -q -x 2>&1 | tail -3 >> $O/parity.txt
(cd /tmp && MIND_HIP_LIB=$GRAFT_REPO_ROOT/mind_amd/libmind_hip_pp.so timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -- python $GRAFT_REPO_ROOT/tests/diag/gpu_diag_predictor.py --prec bf16x3 --timing-only --big > $GRAFT_REPO_ROOT/$O/pmc_sq.log 2>&1)
python tools/pmc_summary.py $O/pmc_sq k_pair > $O/pmc_sq_summary.json 2>> $O/pmc_sq.log
rm -rf $O/pmc_sq
cat $O/ab.txt; cat $O/parity.txt
