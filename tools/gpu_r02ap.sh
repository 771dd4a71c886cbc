#!/bin/bash
# GPU call r02ap: HBM traffic of the pair kernel at the headline size (demo_1, N = 96): two PMC passes (FETCH_SIZE, WRITE_SIZE), nothing else traced
O=gpurun_out/r02ap; mkdir -p $O
export TMPDIR=/tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$cnt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_$cnt.json 2> $GRAFT_REPO_ROOT/$O/err_$cnt.txt)
  python tools/pmc_summary.py $O/pmc_$cnt k_pair_bf > $O/pmc_${cnt}_k_pair_bf.json
  rm -rf $O/pmc_$cnt
  cat $O/pmc_${cnt}_k_pair_bf.json | head -30
done
