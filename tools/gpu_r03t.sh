#!/bin/bash
# GPU call r03t: planner threads in one process after the page-locked staging of the tree-iLQR call
O=gpurun_out/r03t; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/thr.py <<'PY'
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
    [t.start() for t in ths]; [t.join() for t in ths]
    print(sys.argv[1], P, "threads: seconds for 40 plans each", [round(res[i], 3) for i in range(P)], flush=True)
PY
timeout 200 python /tmp/thr.py pinned 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_ilqr.py tests/test_gpu_ilqr_surface.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('demo_1', round(d['value'],1), round(d['ms_per_step'],3))"
