#!/bin/bash
# GPU call r03m: whole GPU suite with the device-built root as the default
O=gpurun_out/r03m; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x -s 2>&1 | tail -25 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
