#!/bin/bash
# GPU call r03j: RCCL on a one-rank nccl group (forced collectives) + the sharded tests
O=gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x -k "rccl or two_sharded" 2>&1 | tail -15 > $O/pytest_sharded.txt; cat $O/pytest_sharded.txt
