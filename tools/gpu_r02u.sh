#!/bin/bash
O=gpurun_out/r02u; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ilqr.py tests/test_gpu_ilqr_surface.py tests/test_gpu_random_sweep.py -m gpu -q -x > $O/pytest_ilqr.txt 2>&1; tail -3 $O/pytest_ilqr.txt
MIND_HIP_LIB=$GRAFT_REPO_ROOT/mind_amd/libmind_hip_prof.so timeout 200 python tools/gpu_ilqr_phase.py demo_1 3 formula_branching:20240121 > $O/ilqr_phase2.txt 2>&1
grep "k_ilqr" $O/ilqr_phase2.txt | sed -n 21,40p | cut -c1-330
