#!/bin/bash
# GPU call r02i: HBM traffic of the bf16x3 pair kernel (two PMC passes), config 5 size in bf16 / bf16x3, new test
O=gpurun_out/r02i; mkdir -p $O
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
(cd /tmp && timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$ctr -- python $GRAFT_REPO_ROOT/tests/diag/gpu_diag_predictor.py --prec bf16x3 --timing-only --big > $GRAFT_REPO_ROOT/$O/pmc_$ctr.log 2>&1)
python tools/pmc_summary.py $O/pmc_$ctr k_pair > $O/pmc_${ctr}_summary.json 2>> $O/pmc_$ctr.log
grep -h "k_pair_bf<1" -r $O/pmc_$ctr --include=*counter_collection.csv | head -400 > $O/pmc_${ctr}_rows.csv
rm -rf $O/pmc_$ctr
done
timeout 300 python bench.py --workload stress128tree --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_stress128tree_bf16x3.json 2> $O/stress_x3.err
MIND_PAIR_PREC=bf16 timeout 300 python bench.py --workload stress128tree --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_stress128tree_bf16.json 2> $O/stress_bf16.err
MIND_PAIR_PREC=bf16 timeout 300 python bench.py --workload cfg4tree --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg4tree_bf16.json 2> $O/cfg4_bf16.err
timeout 300 python -m pytest tests/test_gpu_plan.py -q -x -k "arithmetic" > $O/pytest_arith.txt 2>&1
tail -3 $O/pytest_arith.txt; cat $O/pmc_FETCH_SIZE_summary.json | head -30
