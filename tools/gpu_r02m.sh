#!/bin/bash
# GPU call r02m: MFMA ActorNet (k_actor_mfma): parity tests, A/B against the fp32 VALU kernel (kernel-trace stats)
O=gpurun_out/r02m; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_predictor.py tests/test_gpu_random_sweep.py -m gpu -q -x > $O/pytest_pred.txt 2>&1; echo "rc=$?" >> $O/pytest_pred.txt
tail -4 $O/pytest_pred.txt
for m in 1 0; do
  (cd /tmp && MIND_ENC_MFMA=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt$m -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_demo1_mfma$m.json 2> $GRAFT_REPO_ROOT/$O/bench_demo1_mfma$m.err)
  find $O/kt$m -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_demo_1_mfma$m.csv; rm -rf $O/kt$m
  (cd /tmp && MIND_ENC_MFMA=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kc$m -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4tree --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_cfg4tree_mfma$m.json 2> $GRAFT_REPO_ROOT/$O/bench_cfg4tree_mfma$m.err)
  find $O/kc$m -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_cfg4tree_mfma$m.csv; rm -rf $O/kc$m
  grep -h "k_actor" $O/kernel_stats_demo_1_mfma$m.csv $O/kernel_stats_cfg4tree_mfma$m.csv | cut -c1-160
done
for m in 1 0; do MIND_ENC_MFMA=$m timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('demo_1 mfma=$m', d['value'], d['ms_per_step'])"; done
