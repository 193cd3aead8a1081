#!/bin/bash
# slots per lane with the default bench (4 lanes x calls of 2048): tools/sweep_slots.sh through gpurun
run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --no-cpu-baseline --merged-launches-only $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('$tag', d['value'])"; }
EXTRA=""
for i in 1 2 3; do for s in 384 448 512; do run slots$s GTO_SLOTS=$s; done; done
