run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --no-cpu-baseline --merged-launches-only $EXTRA 2>gpurun_out/e_$tag.err | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); p=d.get('pipeline',{})
print('$tag', d['value'], d['ms_per_step'])"; }
EXTRA="--steps 20 --warmup 5"
for i in 1 2; do run s0 X=1; run s40 GTO_BENCH_STAGGER_US=40; run s80 GTO_BENCH_STAGGER_US=80; run s150 GTO_BENCH_STAGGER_US=150;  run s300 GTO_BENCH_STAGGER_US=300; done
