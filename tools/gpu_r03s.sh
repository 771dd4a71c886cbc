#!/bin/bash
# GPU call r03s: planner threads in one process: hardware-queue count, active wait
O=gpurun_out/r03s; mkdir -p $O
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
timeout 200 python /tmp/thr.py default 2>/dev/null
GPU_MAX_HW_QUEUES=16 timeout 200 python /tmp/thr.py hwq16 2>/dev/null
HIP_FORCE_DEV_KERNARG=1 timeout 200 python /tmp/thr.py devkernarg 2>/dev/null
AMD_DIRECT_DISPATCH=0 timeout 200 python /tmp/thr.py nodirect 2>/dev/null
MIND_NATIVE_AIME=0 timeout 200 python /tmp/thr.py pyrounds 2>/dev/null
