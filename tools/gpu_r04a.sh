#!/bin/bash
# round 4, call a: parity of the tile-native pair kernel + same-box A/B against the row-major one
O=gpurun_out/r04a; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_predictor.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest_predictor.txt; cat $O/pytest_predictor.txt
timeout 600 python tests/diag/gpu_diag_predictor.py --big --timing-only --prec bf16x3,bf16 --tile 0,1 > $O/diag_timing.txt 2>&1; grep -v "^\[" $O/diag_timing.txt | tail -30
bash tools/gpu_ab_tree.sh r04a "MIND_PAIR_TILE=0" "MIND_PAIR_TILE=1"
