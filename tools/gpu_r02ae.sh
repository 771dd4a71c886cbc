#!/bin/bash
O=gpurun_out/r02ae; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_plan.py -m gpu -q -s -k "branching_weights" > $O/pytest_branch.txt 2>&1; grep -E "chosen tree|passed|failed|Error|assert" $O/pytest_branch.txt | head -30
