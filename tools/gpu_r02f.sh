#!/bin/bash
# GPU call r02f: the cleaned-up library: whole GPU suite, bench line, kernel-trace stats of the bench, diag timings
O=gpurun_out/r02f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
timeout 120 python tests/diag/gpu_diag_predictor.py --prec f32,bf16x3,bf16 --timing-only --big 2>&1 | grep -E "timing|arith" > $O/diag_timing.txt
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_kt.json 2> $GRAFT_REPO_ROOT/$O/bench_kt.err)
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_demo_1.csv; rm -rf $O/kt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt2 -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4tree --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_cfg4tree.json 2> $GRAFT_REPO_ROOT/$O/bench_cfg4tree.err)
find $O/kt2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_cfg4tree.csv; rm -rf $O/kt2
tail -3 $O/pytest.txt; cat $O/diag_timing.txt | grep -E "arith|B=24|B=1:"; tail -2 $O/bench.err
