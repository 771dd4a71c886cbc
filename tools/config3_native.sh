# BASELINE config 3 with the native loop (mind_amd/native_loop.py): processes / threads, with and without the loop's speculative warm start, same box
run() { # label, args...
  l="$1"; shift
  timeout 500 python bench.py --workload demo_all "$@" --steps 30 --warmup 3 --no-cpu-baseline --no-extras --no-traffic 2>/tmp/err.txt | tail -1 > /tmp/line.json
  python -c "import json; d=json.loads(open('/tmp/line.json').read()); print('$l', round(d['value'],1), 'sim steps/s', round(d['ms_per_step'],3), 'ms per round of plans')" || tail -3 /tmp/err.txt
}
for rep in 1 2; do
MIND_CONCURRENT_NATIVE=1 MIND_NATIVE_SPECULATE=1 run "x4 native processes, speculative warm start" --concurrent 4 --processes
MIND_CONCURRENT_NATIVE=1 run "x4 native processes" --concurrent 4 --processes
run "x4 python processes (speculative warm start)" --concurrent 4 --processes
MIND_CONCURRENT_NATIVE=1 MIND_NATIVE_SPECULATE=1 run "x4 native threads, speculative warm start" --concurrent 4
MIND_CONCURRENT_NATIVE=1 run "x4 native threads" --concurrent 4
MIND_CONCURRENT_NATIVE=1 MIND_NATIVE_SPECULATE=1 run "x8 native processes, speculative warm start" --concurrent 8 --processes
MIND_CONCURRENT_NATIVE=1 MIND_NATIVE_SPECULATE=1 run "x16 native: 4 processes of 4 threads, speculative" --concurrent 16 --processes --per-process 4
run "x16 python: two processes of event loops" --concurrent 16 --processes --per-process 8
done
MIND_NATIVE_SPECULATE=1 run "x1 native, speculative warm start (headline loop demo_all[0])" 
run "x1 native"
