#!/bin/bash
# A/B of prebuilt libraries (tools/ab_build.sh -> tools/_ab/lib_<tag>.so) on the four workloads of the bench line:
# tools/ab_run.sh <reps> tag1 tag2 ...   prints trajectories/s: default run | the driver's call | configs[2] | configs[4]
reps=$1; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
val() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']/1e3,1), end=' ')
except Exception as e: print('ERR', end=' ')"; }
for rep in $(seq 1 $reps); do for t in "$@"; do
  export GTO_HIP_LIB=$PWD/tools/_ab/lib_$t.so
  echo -n "$t: default "; python bench.py --no-cpu-baseline --no-other-configs --no-next-rows --merged-launches-only 2>/dev/null | val
  echo -n "| steps20 "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-next-rows --merged-launches-only 2>/dev/null | val
  echo -n "| cfg2 "; python bench.py --light --no-cpu-baseline --no-next-rows --no-other-configs --repeats 3 --warmup 1 --robot fetch --batch 256 --shelf --merge 8 --steps 32 2>/dev/null | val
  echo -n "| cfg4 "; python bench.py --light --no-cpu-baseline --no-next-rows --no-other-configs --repeats 3 --warmup 1 --robot fetch_mobile --T 80 --grid 256 --shelf --batch 64 --merge 8 --steps 32 2>/dev/null | val
  echo; done; done
