# Developer tool (GPU box): the official K = 20 region with the W warm-up steps in front of / behind the long untimed stretch (MTD_BENCH_WARM_ORDER)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3 4 5 6 7 8; do
for o in 0 1; do
  MTD_BENCH_WARM_ORDER=$o timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['extra'].get('timed_region_repeats',{}).get('ms_per_step',[])
print('warm-up first=$o: official %.2f us  repeats median %.2f min %.2f' % (1e3*d['ms_per_step'], 1e3*sorted(r)[len(r)//2], 1e3*min(r)))"
done; done
