#!/bin/bash
O=gpurun_out/r02o; mkdir -p $O
timeout 300 python tools/gpu_cprofile.py demo_1 40 formula_branching:20240121 > $O/cprofile_demo_1.txt 2>&1
timeout 200 python tools/gpu_time_host.py demo_1 40 formula_branching:20240121 > $O/host_time_demo_1.txt 2>&1
tail -30 $O/host_time_demo_1.txt
