run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --no-cpu-baseline $EXTRA 2>gpurun_out/e_$tag.err | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); p=d.get('pipeline',{})
print('$tag', d['value'], p.get('serial_trajectories_per_s'), d['host_api_pipelined_trajectories_per_s'], d['host_api_trajectories_per_s'])"; }
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
EXTRA="--steps 20 --warmup 5"
for i in 1 2 3 4 5; do run s20 X=1; done
EXTRA=""
run def X=1; run def X=1
python tools/planner_latency.py panda_5k 64 30 2>&1 | tail -2
