#!/bin/bash
# one PMC pass of tools/dbg_run.py: usage: B=2048 tools/pmc_one.sh <outdir> COUNTER...
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/one -o p -- python $GRAFT_REPO_ROOT/tools/dbg_run.py > $out/one.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out
python - <<PY
import csv,glob
for f in glob.glob("$out/one/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "traj" in r["Kernel_Name"]: print("kernel ns", int(r["End_Timestamp"])-int(r["Start_Timestamp"]), "LDS", r.get("LDS_Block_Size"), "scratch", r.get("Scratch_Size"), "vgpr", r.get("VGPR_Count"))
PY
