#!/bin/bash
# GPU call r02c: pair kernel variants (LDS waits, direct edge moves), parity of the winner, new tests, host profile, fused vs processes
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
for v in w1d0 w0d0 w0d1 w0d1p1 w1d1; do
  echo "== variant $v" >> $O/ab.txt
  MIND_HIP_LIB=$PWD/mind_amd/libmind_hip_$v.so timeout 120 python tests/diag/gpu_diag_predictor.py --prec bf16x3 --timing-only --big 2>&1 | grep timing >> $O/ab.txt
done
for v in w0d0 w0d1; do
  echo "== parity of variant $v" >> $O/parity.txt
  MIND_HIP_LIB=$PWD/mind_amd/libmind_hip_$v.so timeout 200 python -m pytest tests/test_gpu_predictor.py -q -x 2>&1 | tail -3 >> $O/parity.txt
done
timeout 500 python -m pytest tests/test_gpu_plan.py -q -x -s -k "branching or fused" > $O/pytest_new.txt 2>&1
timeout 120 python tools/gpu_time_host.py demo_1 20 > $O/host_demo_1.txt 2>&1
timeout 120 python tools/gpu_time_host.py demo_1 20 formula_branching:20240121 > $O/host_demo_1_branching.txt 2>&1
for ck in "" "--ckpt formula_branching:20240121"; do
  timeout 200 python bench.py --workload demo_all --concurrent 4 --fused --steps 20 --warmup 2 $ck >> $O/fused.json 2>> $O/fused.err
  timeout 200 python bench.py --workload demo_all --concurrent 4 --processes --steps 20 --warmup 2 $ck >> $O/procs.json 2>> $O/procs.err
done
timeout 200 python bench.py --workload demo_1 --ckpt formula_branching:20240121 --steps 60 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_branching.json 2> $O/bench_branching.err
tail -5 $O/ab.txt; cat $O/parity.txt; tail -4 $O/pytest_new.txt
