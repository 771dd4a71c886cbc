#!/bin/bash
# GPU call r02t: suite after making the decoder MFMA kernel opt-in; default bench line; cfg4tree bench
O=gpurun_out/r02t; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('demo_1', d['value'], d['ms_per_step'], d['breakdown_ms'], d['roofline']['frac'], d['cpu_baseline'])"
timeout 300 python bench.py --workload cfg4tree --no-cpu-baseline --no-extras > $O/bench_cfg4tree.json 2> $O/bench_cfg4tree.err
python -c "import json; d=json.loads(open('$O/bench_cfg4tree.json').read().strip().splitlines()[-1]); print('cfg4tree', d['value'], d['ms_per_step'], d['nodes_expanded_per_s'], d['roofline']['frac'])"
