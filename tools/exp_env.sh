run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --no-cpu-baseline $EXTRA 2>gpurun_out/e_$tag.err | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); p=d.get('pipeline',{})
print('$tag', d['value'], p.get('lane_results_reproducible_alone'), p.get('merged_equals_single_batch_solves'), p.get('serial_trajectories_per_s'), d['roofline']['avg_launch_us'])"; }
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
EXTRA="--steps 20 --warmup 5"
for i in 1 2 3; do run nw4 GTO_STEP_NW_FEW=4; run nw8 GTO_STEP_NW_FEW=8; done
EXTRA=""
run def_nw4 GTO_STEP_NW_FEW=4; run def_nw8 GTO_STEP_NW_FEW=8
