#!/bin/bash
# GPU call r02b: A/B of the bf16 pair kernel's load placement, phase trace, SQ counters, new parity tests, kernel stats of the bench
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
for v in p0q0 p1q0 p0q1 p1q1; do
  echo "== variant $v" >> $O/ab.txt
  MIND_HIP_LIB=$PWD/mind_amd/libmind_hip_$v.so timeout 120 python tests/diag/gpu_diag_predictor.py --prec bf16x3 --timing-only --big 2>&1 | grep timing >> $O/ab.txt
done
MIND_HIP_LIB=$PWD/mind_amd/libmind_hip_trace.so timeout 120 python tests/diag/gpu_diag_predictor.py --prec bf16x3 --timing-only 2>&1 | grep -E "k_pair_bf|timing" | tail -40 > $O/trace.txt
(cd /tmp && MIND_HIP_LIB=$GRAFT_REPO_ROOT/mind_amd/libmind_hip_p0q0.so timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -- python $GRAFT_REPO_ROOT/tests/diag/gpu_diag_predictor.py --prec bf16x3 --timing-only --big > $GRAFT_REPO_ROOT/$O/pmc_sq.log 2>&1)
python tools/pmc_summary.py $O/pmc_sq k_pair > $O/pmc_sq_summary.json 2>> $O/pmc_sq.log
(cd /tmp && MIND_HIP_LIB=$GRAFT_REPO_ROOT/mind_amd/libmind_hip_p0q0.so timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq2 -- python $GRAFT_REPO_ROOT/tests/diag/gpu_diag_predictor.py --prec bf16x3 --timing-only --big > $GRAFT_REPO_ROOT/$O/pmc_sq2.log 2>&1)
python tools/pmc_summary.py $O/pmc_sq2 k_pair > $O/pmc_sq2_summary.json 2>> $O/pmc_sq2.log
rm -rf $O/pmc_sq $O/pmc_sq2
timeout 400 python -m pytest tests/test_gpu_plan.py -q -x -k "branching or checkpoint or bench_prints" > $O/pytest_new.txt 2>&1
timeout 500 python -m pytest tests/test_gpu_sharded.py -q -x > $O/pytest_sharded.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_kt.json 2> $GRAFT_REPO_ROOT/$O/bench_kt.err)
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/kt
tail -3 $O/ab.txt; tail -3 $O/pytest_new.txt; tail -3 $O/pytest_sharded.txt
