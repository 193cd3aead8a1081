#!/bin/bash
# A/B: waypoints per workgroup of the obstacle kernel, in the default bench regime (4 lanes x calls of 32 steps)
cd $GRAFT_REPO_ROOT
for g in 1 2 3 4 6 8; do echo -n "TG=$g: "; GTO_OBS_TG=$g python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['pipeline']['serial_trajectories_per_s'], d['iters_mean'])"; done
