#!/bin/bash
O=gpurun_out/r02j; mkdir -p $O
export TMPDIR=/tmp
for x in 0 1; do
  echo "== MIND_XCD_ORDER=$x" >> $O/ab.txt
  MIND_XCD_ORDER=$x timeout 120 python tests/diag/gpu_diag_predictor.py --prec bf16x3 --timing-only --big 2>&1 | grep timing >> $O/ab.txt
done
(cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_f -- python $GRAFT_REPO_ROOT/tests/diag/gpu_diag_predictor.py --prec bf16x3 --timing-only --big > $GRAFT_REPO_ROOT/$O/pmc_f.log 2>&1)
grep -h "k_pair_bf<1" -r $O/pmc_f --include=*counter_collection.csv | head -400 > $O/pmc_FETCH_SIZE_rows.csv; rm -rf $O/pmc_f
timeout 300 python -m pytest tests/test_gpu_predictor.py tests/test_gpu_plan.py -q -x -k "arithmetic or predictor" > $O/pytest.txt 2>&1
timeout 300 python bench.py --workload cfg4tree --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg4tree.json 2> $O/cfg4.err
cat $O/ab.txt; tail -3 $O/pytest.txt
