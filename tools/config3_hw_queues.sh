export MIND_CONCURRENT_NATIVE=1
run() { # label, args...
  l="$1"; shift
  timeout 500 python bench.py --workload demo_all "$@" --steps 30 --warmup 3 --no-cpu-baseline --no-extras --no-traffic 2>/tmp/err.txt | tail -1 > /tmp/line.json
  python -c "import json; d=json.loads(open('/tmp/line.json').read()); print('$l', round(d['value'],1), 'sim steps/s', round(d['ms_per_step'],3), 'ms per round of plans')" || tail -3 /tmp/err.txt
}
for q in 4 8 16 24; do
  GPU_MAX_HW_QUEUES=$q run "x4 native threads, GPU_MAX_HW_QUEUES=$q" --concurrent 4
  GPU_MAX_HW_QUEUES=$q run "x16 native threads, GPU_MAX_HW_QUEUES=$q" --concurrent 16
  GPU_MAX_HW_QUEUES=$q run "x16 native: 2 processes of 8 threads, GPU_MAX_HW_QUEUES=$q" --concurrent 16 --processes --per-process 8
  GPU_MAX_HW_QUEUES=$q MIND_CONCURRENT_NATIVE=0 run "x16 python: 2 processes of event loops, GPU_MAX_HW_QUEUES=$q" --concurrent 16 --processes --per-process 8
done
