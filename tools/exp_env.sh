#!/bin/bash
# A/B harness for environment knobs (INTEGRATION.md "Environment knobs"): runs bench.py under each setting and prints
# value / one-batch-at-a-time rate / host-API rates.  Edit the `run` lines; through gpurun: bash tools/exp_env.sh
run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --no-cpu-baseline $EXTRA 2>gpurun_out/e_$tag.err | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); p=d.get('pipeline',{})
print('$tag', d['value'], p.get('serial_trajectories_per_s'), d['host_api_pipelined_trajectories_per_s'], d['host_api_trajectories_per_s'], d['quality']['gate'])"; }
mkdir -p gpurun_out
EXTRA="--steps 20 --warmup 5"
run default X=1
run tg3_everywhere GTO_OBS_TG=3
run no_interleave GTO_OBS_INTERLEAVE=0
run step_nw4 GTO_STEP_NW_FEW=4
