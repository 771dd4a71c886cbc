#!/bin/bash
# what the exchanges of the sharded native plan cost on ONE rank: the cfg4 full tree planned by a plain process and by a one-rank nccl
# group whose collectives really run (MIND_FORCE_COLLECTIVES=1: RCCL on this GPU, every pack / all-gather / unpack / all-reduce executed)
O=gpurun_out/${1:-shard_overhead}; mkdir -p $O
for mode in plain forced; do
  if [ $mode = forced ]; then X="MIND_FORCE_COLLECTIVES=1 MIND_DIST_BACKEND=nccl"; else X=""; fi
  env $X RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29591 MIND_TEST_WORKLOAD=cfg4tree python tests/dist_gpu_worker.py $O/$mode.pkl 8 > $O/$mode.log 2>&1
  python - <<PY
import pickle
d = pickle.load(open("$O/$mode.pkl", "rb"))
w = d["wall_ms"][2:]
print("$mode: ms per plan", w, "mean", round(sum(w) / len(w), 2), "| collectives", d["collectives"], "gathered MB", round(d["gathered"] / 1e6, 1), "| aime ms/plan", round(d["timing"]["aime_s"] / d["timing"]["plans"] * 1e3, 2))
PY
done | tee $O/summary.txt
