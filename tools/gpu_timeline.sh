#!/bin/bash
# GPU call ${1:-timeline}: GPU timeline of one planning cycle of the headline loop (kernel + copy trace)
O=gpurun_out/${1:-timeline}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err
cd $GRAFT_REPO_ROOT && python tools/gpu_timeline.py $O/trace > $O/timeline.txt 2>&1; cat $O/timeline.txt | tail -90
find $O/trace -name "*.csv" -size +2M -delete
