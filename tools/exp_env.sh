run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --no-cpu-baseline $EXTRA 2>gpurun_out/e_$tag.err | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); p=d.get('pipeline',{})
print('$tag', d['value'], p.get('lane_results_reproducible_alone'), p.get('merged_equals_single_batch_solves'), p.get('serial_trajectories_per_s'))"; }
EXTRA="--mode single"
run single X=1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mode or single or 1-" 2>&1 | tail -3
