#!/bin/bash
# SQ counters of k_ilqr on the headline loop (one --pmc pass + kernel trace): what the serial loops spend their cycles on
O=gpurun_out/${1:-pmc_ilqr}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/sq -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-traffic > /dev/null 2>&1
cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/inst -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-traffic > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/sq k_ilqr > $O/pmc_sq_k_ilqr.json 2>&1
python tools/pmc_summary.py $O/inst k_ilqr > $O/pmc_inst_k_ilqr.json 2>&1
rm -rf $O/sq $O/inst
python - <<PY
import json
for f in ("$O/pmc_sq_k_ilqr.json", "$O/pmc_inst_k_ilqr.json"):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); print(open(f).read()[:600]); continue
    for k, v in d.items():
        print(k, "avg_us", round(v.get("avg_us", 0), 1), {c: round(x["per_dispatch"]) for c, x in v["counters"].items()})
PY
