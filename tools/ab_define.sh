#!/bin/bash
# A/B of a compile-time knob of the kernels: tools/ab_define.sh NAME v1 v2 ...   (default bench regime)
set -e
name=$1; shift
cd $GRAFT_REPO_ROOT/grasptrajopt_amd/csrc
for v in "$@"; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -D$name=$v gto_api.hip -o /tmp/lib_$v.so 2>/dev/null; done
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "$@"; do echo -n "$name=$v: "; GTO_HIP_LIB=/tmp/lib_$v.so python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['pipeline']['serial_trajectories_per_s'], d['pipeline']['merged_equals_single_batch_solves'])"; done; done
