#!/bin/bash
# ONE rocprofv3 PMC pass of chosen counters on bench.py (one lane, calls of the timed region's size), per-kernel means.
# usage: tools/pmc_one.sh <outdir-under-gpurun_out> <steps> <merge> COUNTER [COUNTER ...]      (<= 8 SQ counters per pass)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; steps=$2; merge=$3; shift 3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/one -o p -- python $GRAFT_REPO_ROOT/bench.py --pipeline 1 --steps $steps --merge $merge --warmup 1 --no-cpu-baseline --merged-launches-only --repeats 1 > $out/one.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out | grep -v "^launches" 
rm -rf $out/one
