#!/bin/bash
# GPU call r02af: final verification of the round: whole GPU suite, smoke, default bench
O=gpurun_out/r02af; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('demo_1', d['value'], d['ms_per_step'], d['breakdown_ms']['aime'], d['breakdown_ms']['ilqr'], 'frac', d['roofline']['frac']); print('tree', d['tree']['ms_per_plan'], d['tree']['nodes_expanded_per_s'])"
