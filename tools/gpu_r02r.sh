#!/bin/bash
# GPU call r02r: decoder actor part on the MFMA (k_dec_actor_mfma): suite + kernel-trace stats
O=gpurun_out/r02r; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -6 $O/pytest.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_demo1.json 2> $GRAFT_REPO_ROOT/$O/bench_demo1.err)
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_demo_1.csv; rm -rf $O/kt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kc -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4tree --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_cfg4tree.json 2> $GRAFT_REPO_ROOT/$O/bench_cfg4tree.err)
find $O/kc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_cfg4tree.csv; rm -rf $O/kc
grep -h "k_dec\|k_aime_select" $O/kernel_stats_demo_1.csv $O/kernel_stats_cfg4tree.csv | cut -c1-170
