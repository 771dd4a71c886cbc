#!/bin/bash
# GPU call r03r: why do P planner threads in one process not overlap?  speculative side contexts, GIL switch interval
O=gpurun_out/r03r; mkdir -p $O
export TMPDIR=/tmp
run() { timeout 400 python bench.py --workload demo_all --concurrent 4 $2 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-traffic 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'sim steps/s')" || tail -3 $O/err.txt; }
run "threads default"
MIND_SPECULATIVE_WARM_START=0 run "threads spec=0"
MIND_SWITCH_INTERVAL=0.0001 run "threads switch=1e-4"
MIND_SWITCH_INTERVAL=0.0001 MIND_SPECULATIVE_WARM_START=0 run "threads spec=0 switch=1e-4"
MIND_SPECULATIVE_WARM_START=0 run "processes spec=0" --processes
python - <<'PY'
# two threads calling the native plan concurrently: do the C calls overlap?
import sys, time, threading; sys.path.insert(0, '.')
import torch
from bench import WORKLOADS, make_closed_loop
res = {}
def worker(i, n, bar):
    with torch.cuda.stream(torch.cuda.Stream()):
        pl, sim, w = make_closed_loop(dict(WORKLOADS["demo_%d" % (i % 4 + 1)]), speculative=False)
        sim.run_plans(3)
        bar.wait()
        t0 = time.perf_counter(); sim.run_plans(n); torch.cuda.current_stream().synchronize()
        res[i] = time.perf_counter() - t0
for P in (1, 2, 4):
    bar = threading.Barrier(P); ths = [threading.Thread(target=worker, args=(i, 40, bar)) for i in range(P)]
    t0 = time.perf_counter(); [t.start() for t in ths]; [t.join() for t in ths]
    print(P, "threads: per-thread seconds for 40 plans", [round(res[i], 3) for i in range(P)])
PY
