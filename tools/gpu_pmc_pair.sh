#!/bin/bash
# SQ counters of the pair kernel on the cfg4 full tree (one --pmc pass, kernel trace only beside it): MFMA-pipe busy, wave cycles, waits, LDS
O=gpurun_out/${1:-pmc_pair}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/sq -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4tree --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-traffic > /dev/null 2>&1
cd /tmp && rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/lds -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4tree --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-traffic > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/sq k_pair_bf > $O/pmc_sq_k_pair_bf.json 2>&1
python tools/pmc_summary.py $O/lds k_pair_bf > $O/pmc_lds_k_pair_bf.json 2>&1
rm -rf $O/sq $O/lds
head -c 1500 $O/pmc_sq_k_pair_bf.json; echo; head -c 1500 $O/pmc_lds_k_pair_bf.json
