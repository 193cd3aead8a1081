run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --no-cpu-baseline $EXTRA 2>gpurun_out/e_$tag.err | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); p=d.get('pipeline',{})
print('$tag', d['value'], p.get('serial_trajectories_per_s'), d['roofline']['avg_launch_us'])"; }
EXTRA="--steps 20 --warmup 5"
for i in 1 2 3; do run new20 X=1; run old20 GTO_HIP_LIB=$PWD/tools/ab_old.so; done
EXTRA=""
for i in 1 2; do run newdef X=1; run olddef GTO_HIP_LIB=$PWD/tools/ab_old.so; done
