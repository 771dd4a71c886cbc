#!/bin/bash
# GPU call r02s: k_aime_select rewrite, decoder MFMA threshold + forced test, memory-copy trace of the demo_1 loop
O=gpurun_out/r02s; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -6 $O/pytest.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_demo1.json 2> $GRAFT_REPO_ROOT/$O/bench_demo1.err)
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_demo_1.csv
find $O/kt -name "*memory_copy_stats.csv" | head -1 | xargs -I{} cp {} $O/memory_copy_stats_demo_1.csv
find $O/kt -name "*memory_copy_trace.csv" | head -1 | xargs -I{} cp {} $O/memory_copy_trace_demo_1.csv
ls $O/kt/* | head; rm -rf $O/kt
grep -h "k_dec\|k_aime_select" $O/kernel_stats_demo_1.csv | cut -c1-170
cat $O/memory_copy_stats_demo_1.csv | head
