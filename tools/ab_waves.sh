#!/bin/bash
# A/B: register budget of the obstacle kernel (waves per SIMD the allocator leaves room for) x list capacity x TG
set -e
cd $GRAFT_REPO_ROOT/grasptrajopt_amd/csrc
for cfg in "4 64" "5 64" "5 48" "6 48"; do set -- $cfg; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DGTO_OBS_MIN_WAVES=$1 -DGTO_LIST_CAP=$2 gto_api.hip -o /tmp/lib_w$1_$2.so -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "k_obstacle_gram" | grep -i "VGPRs:\|Scratch\|Occupancy" | tr '\n' ' '; echo " <- $cfg"; done
cd $GRAFT_REPO_ROOT
for tg in 2 3; do for cfg in "4 64" "5 64" "5 48" "6 48"; do set -- $cfg; echo -n "TG=$tg minwaves=$1 cap=$2: "; GTO_OBS_TG=$tg GTO_HIP_LIB=/tmp/lib_w$1_$2.so python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['pipeline']['serial_trajectories_per_s'])"; done; done
