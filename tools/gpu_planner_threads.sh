#!/bin/bash
# GPU call r03u: where do two planner threads in one process lose their time?
export TMPDIR=/tmp
cat > /tmp/thr2.py <<'PY'
import sys, time, threading, os; sys.path.insert(0, '.')
import torch
from bench import WORKLOADS, make_closed_loop
res = {}
def worker(i, n, bar):
    with torch.cuda.stream(torch.cuda.Stream()):
        pl, sim, w = make_closed_loop(dict(WORKLOADS["demo_%d" % (i % 4 + 1)]), speculative=False)
        sim.run_plans(3)
        bar.wait()
        s0 = dict(pl.timing_sum)
        t0 = time.perf_counter(); sim.run_plans(n); torch.cuda.current_stream().synchronize()
        res[i] = (time.perf_counter() - t0, {k: (pl.timing_sum[k] - s0[k]) / n * 1e3 for k in ("aime_s", "ilqr_s", "total_s")})
for P in (1, 2):
    bar = threading.Barrier(P); ths = [threading.Thread(target=worker, args=(i, 40, bar)) for i in range(P)]
    [t.start() for t in ths]; [t.join() for t in ths]
    for i in range(P): print(P, "threads, thread", i, round(res[i][0], 3), "s;", {k: round(v, 2) for k, v in res[i][1].items()}, flush=True)
PY
timeout 200 python /tmp/thr2.py 2>/dev/null
