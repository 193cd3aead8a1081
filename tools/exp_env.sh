run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --no-cpu-baseline $EXTRA 2>gpurun_out/e_$tag.err | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); p=d.get('pipeline',{})
print('$tag', d['value'], p.get('lane_results_reproducible_alone'), p.get('merged_equals_single_batch_solves'), p.get('serial_trajectories_per_s'), d['roofline']['avg_launch_us'], d['quality']['gate'])"; }
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
GTO_DEBUG_TIMING=1 B=64 REPS=2 python tools/dbg_run.py 2>&1 | grep "step-kernel" | tail -1
EXTRA="--steps 20 --warmup 5"
for i in 1 2 3; do run s20 X=1; done
EXTRA=""
run def X=1; run def X=1
