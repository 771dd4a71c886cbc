#!/bin/bash
O=gpurun_out/r02ah; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_plan.py -m gpu -q -s -k "branching_weights_whole_run" > $O/pytest_branch_runs.txt 2>&1; grep -E "cycles with|passed|failed|Error|assert " $O/pytest_branch_runs.txt | head -30
