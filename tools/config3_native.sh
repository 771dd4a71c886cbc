# BASELINE config 3 with the native loop (mind_amd/native_loop.py): threads / processes, same box.
export MIND_CONCURRENT_NATIVE=1
run() { # label, args...
  l="$1"; shift
  timeout 500 python bench.py --workload demo_all "$@" --steps 30 --warmup 3 --no-cpu-baseline --no-extras --no-traffic 2>/tmp/err.txt | tail -1 > /tmp/line.json
  python -c "import json; d=json.loads(open('/tmp/line.json').read()); print('$l', round(d['value'],1), 'sim steps/s', round(d['ms_per_step'],3), 'ms per round of plans')" || tail -3 /tmp/err.txt
}
for rep in 1 2; do
for Q in 1 2 4 8 16; do run "x16 native: processes of $Q threads" --concurrent 16 --processes --per-process $Q; done
MIND_CONCURRENT_NATIVE=0 run "x16 two processes of event loops (python)" --concurrent 16 --processes --per-process 8
run "x8 native: 4 processes of 2 threads" --concurrent 8 --processes --per-process 2
run "x8 native: 2 processes of 4 threads" --concurrent 8 --processes --per-process 4
run "x32 native: 4 processes of 8 threads" --concurrent 32 --processes --per-process 8
done
