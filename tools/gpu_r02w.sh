#!/bin/bash
O=gpurun_out/r02w; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_aime_world.py tests/test_gpu_aime_golden.py tests/test_gpu_plan.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 200 python tools/gpu_time_host.py demo_1 40 formula_branching:20240121 > $O/host_time_demo_1.txt 2>&1
tail -45 $O/host_time_demo_1.txt
