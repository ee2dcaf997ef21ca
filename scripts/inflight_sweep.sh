for n in 2 3 4 2; do
  python bench.py --steps 12 --warmup 3 --cpu-baseline off --no-roofline --inflight $n 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('inflight', $n, round(d['value'],1), 'maps/s', round(d['ms_per_step'],2), 'ms')"
done
