#!/bin/bash
# per-kernel times of the headline loop: rocprofv3 --kernel-trace --stats over a short bench run
O=gpurun_out/${1:-kstats}; mkdir -p $O
export TMPDIR=/tmp
shift
cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-traffic > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; cut -c1-150 $O/kernel_stats.csv | head -16
find $O/trace -name "*.csv" -size +1M -delete
