run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --no-cpu-baseline $EXTRA 2>gpurun_out/e_$tag.err | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); p=d.get('pipeline',{})
print('$tag', d['value'], p.get('lane_results_reproducible_alone'), p.get('merged_equals_single_batch_solves'), p.get('serial_trajectories_per_s'), d['roofline']['avg_launch_us'], p.get('host_cpu_cores_busy'))"; }
EXTRA=""
run def4 X=1; run def8 GTO_CHECK_EVERY=8; run def16 GTO_CHECK_EVERY=16
EXTRA="--steps 20 --warmup 5"
for i in 1 2; do run s20_4 X=1; run s20_8 GTO_CHECK_EVERY=8; run s20_16 GTO_CHECK_EVERY=16; run s20_6 GTO_CHECK_EVERY=6; done
