# BASELINE config 3 drivers under GPU_MAX_HW_QUEUES (hardware queues the HIP runtime maps a process's streams onto): same box, two repetitions
run() { # label, args...
  l="$1"; shift
  timeout 500 python bench.py --workload demo_all "$@" --steps 30 --warmup 3 --no-cpu-baseline --no-extras --no-traffic 2>/tmp/err.txt | tail -1 > /tmp/line.json
  python -c "import json; d=json.loads(open('/tmp/line.json').read()); print('$l', round(d['value'],1), 'sim steps/s', round(d['ms_per_step'],3), 'ms per round of plans')" || tail -3 /tmp/err.txt
}
for rep in 1 2; do
for q in unset 4 8 12; do
  if [ $q = unset ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  MIND_CONCURRENT_NATIVE=1 run "x4 native threads, GPU_MAX_HW_QUEUES=$q" --concurrent 4
  run "x4 python one-thread event loop, GPU_MAX_HW_QUEUES=$q" --concurrent 4 --pipelined
  run "x4 python processes, GPU_MAX_HW_QUEUES=$q" --concurrent 4 --processes
  run "x16 python two processes of event loops, GPU_MAX_HW_QUEUES=$q" --concurrent 16 --processes --per-process 8
done
done
