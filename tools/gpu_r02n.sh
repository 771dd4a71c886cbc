#!/bin/bash
# GPU call r02n: whole GPU suite with the MFMA ActorNet (three-way split) as the default, kernel-trace A/B of the ActorNet kernels
O=gpurun_out/r02n; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -8 $O/pytest.txt
for m in 6 3 0; do
  export MIND_ACTOR_SPLIT=$m MIND_ENC_MFMA=1; if [ $m = 0 ]; then export MIND_ENC_MFMA=0; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt$m -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_demo1_split$m.json 2> $GRAFT_REPO_ROOT/$O/bench_demo1_split$m.err)
  find $O/kt$m -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_demo_1_split$m.csv; rm -rf $O/kt$m
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kc$m -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4tree --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_cfg4tree_split$m.json 2> $GRAFT_REPO_ROOT/$O/bench_cfg4tree_split$m.err)
  find $O/kc$m -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_cfg4tree_split$m.csv; rm -rf $O/kc$m
  grep -h "k_actor" $O/kernel_stats_demo_1_split$m.csv $O/kernel_stats_cfg4tree_split$m.csv | cut -c1-160
done
