#!/bin/bash
# GPU call r02ac: smoke(), default bench line, kernel trace of the same command
O=gpurun_out/r02ac; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err; tail -1 $O/bench.err
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('demo_1', d['value'], d['ms_per_step'], d['breakdown_ms'], 'frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value']); print('tree', d['tree']['ms_per_plan'], d['tree']['nodes_expanded_per_s']); print({k: (v.get('sim_steps_per_s') if isinstance(v, dict) else v) for k, v in d['recorded_scenes'].items()}); print(d['plain_formula_weights'])"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_kt.json 2> $GRAFT_REPO_ROOT/$O/bench_kt.err)
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_demo_1.csv; rm -rf $O/kt
head -12 $O/kernel_stats_demo_1.csv | cut -c1-120
