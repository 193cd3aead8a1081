run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --no-cpu-baseline $EXTRA 2>gpurun_out/e_$tag.err | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); p=d.get('pipeline',{})
print('$tag', d['value'], p.get('serial_trajectories_per_s'))"; }
EXTRA="--steps 20 --warmup 5"
for i in 1 2; do run few64 X=1; run few96 GTO_FEW_INSTANCES=96; run few128 GTO_FEW_INSTANCES=128; run few192 GTO_FEW_INSTANCES=192; run few32 GTO_FEW_INSTANCES=32; done
