#!/bin/bash
# Experiment: obstacle kernel compiled WITHOUT the goal-term code (results are wrong: for occupancy timing only)
set -e
cd $GRAFT_REPO_ROOT/grasptrajopt_amd/csrc
sed 's|      trial_goal_terms_wave(rb, bp, sp, B, bg, tid, 1 - bp.state\[bg\].slot, bp.state + bg, s_q2, s_fr2, s_ga, s_gs);|#ifndef GTO_NO_GOAL_WG\n&\n#endif|' gto_kernels.h > /tmp/k.h
cp gto_kernels.h /tmp/k_orig.h; cp /tmp/k.h gto_kernels.h
for w in 5 6; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DGTO_NO_GOAL_WG -DGTO_OBS_MIN_WAVES=$w gto_api.hip -o /tmp/lib_ng$w.so 2>/dev/null; done
cp /tmp/k_orig.h gto_kernels.h
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for w in 5 6; do for tg in 2 3; do echo -n "nogoal minwaves=$w TG=$tg: "; GTO_OBS_TG=$tg GTO_HIP_LIB=/tmp/lib_ng$w.so python bench.py --no-cpu-baseline --merged-launches-only --max-iter 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['iters_mean'])"; done; done; done
