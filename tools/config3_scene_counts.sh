for cfg in "4 10" "4 4" "16 10" "16 2" "16 1" "64 1"; do
  set -- $cfg; P=$1; S=$2
  for mode in "--pipelined" "--processes"; do
    [ "$P" = 64 ] && [ "$mode" = "--processes" ] && continue
    MIND_ILQR_SLOTS=$S timeout 500 python bench.py --workload demo_all --concurrent $P $mode --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-traffic 2>/tmp/err.txt | tail -1 > /tmp/line.json
    python -c "import json; d=json.loads(open('/tmp/line.json').read()); print('demo_all x$P slots $S [$mode]', round(d['value'],1), 'sim steps/s', round(d['ms_per_step'],3), 'ms per round of plans')" || tail -3 /tmp/err.txt
  done
done
