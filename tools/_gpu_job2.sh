#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/grasptrajopt_amd/csrc
for rep in 1 2; do
for lib in vC vA vB r03; do
  if [ $lib = vC ]; then unset GTO_HIP_LIB; else export GTO_HIP_LIB=$L/libgto_hip_$lib.so; fi
  echo "== $lib $(B=1,64 REPS=40 timeout 300 python tools/serial_latency.py 2>&1 | tail -2 | cut -c1-40 | tr '\n' ' ')"
done; done
for lib in vC vA vB; do
  if [ $lib = vC ]; then unset GTO_HIP_LIB; else export GTO_HIP_LIB=$L/libgto_hip_$lib.so; fi
  for a in "--steps 20 --warmup 5" ""; do timeout 600 python bench.py $a --no-cpu-baseline --no-next-rows 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
print('$lib $a', d['value'], d['timed_regions']['ms_per_step_all'], d['quality']['gate'], 'serial', d['pipeline']['serial_trajectories_per_s'])"; done
done
