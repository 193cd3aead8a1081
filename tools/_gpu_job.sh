cd $GRAFT_REPO_ROOT
for w in 1 2 3 2 3; do GTO_BENCH_HOST_WARM=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline --repeats 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
print($w, d['value'], d['timed_regions']['ms_per_step_all'], '| host', d['host_api']['trajectories_per_s'], d['host_api']['ms_per_step_all'])"; done
