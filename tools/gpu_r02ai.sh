#!/bin/bash
# stress: the multi-workgroup tree-iLQR tests 15 times, the full-tree plan test, then the suite twice
O=gpurun_out/r02ai; mkdir -p $O
for i in $(seq 1 15); do timeout 120 python -m pytest tests/test_gpu_ilqr.py -m gpu -q -x -k "wide" 2>&1 | tail -1; done | sort | uniq -c
for i in 1 2; do timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -1; done
