#!/bin/bash
O=gpurun_out/r02l; mkdir -p $O
for x in 0 1; do
  echo "== MIND_PAIR_TWO_TILE=$x" >> $O/ab.txt
  MIND_PAIR_TWO_TILE=$x timeout 120 python tests/diag/gpu_diag_predictor.py --prec bf16x3,bf16 --timing-only --big 2>&1 | grep -E "timing|arith" >> $O/ab.txt
done
timeout 400 python -m pytest tests/test_gpu_predictor.py tests/test_gpu_random_sweep.py -q -x > $O/pytest.txt 2>&1
cat $O/ab.txt | grep -E "TWO|arith|B=4|B=24"; tail -5 $O/pytest.txt
