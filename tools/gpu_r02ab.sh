#!/bin/bash
O=gpurun_out/r02ab; mkdir -p $O
timeout 300 python tools/gpu_time_host.py cfg4tree 3 > $O/host_time_cfg4tree.txt 2>&1
tail -42 $O/host_time_cfg4tree.txt
