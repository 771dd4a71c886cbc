#!/bin/bash
# per-kernel times of the cfg4 full-tree plan: tools/gpu_kstats_tree.sh <tag> [ENV=..]
O=gpurun_out/${1:-kstats_tree}; mkdir -p $O
export TMPDIR=/tmp
shift
cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4tree --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-traffic > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; cut -c1-60,150-260 $O/kernel_stats.csv | head -12
rm -rf $O/trace
