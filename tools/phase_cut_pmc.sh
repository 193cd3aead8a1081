#!/bin/bash
# Instructions and cycles of k_obstacle_gram by phase: tools/phase_cut.py under rocprofv3 PMC passes, one per cut.
# usage (on the GPU box): bash tools/phase_cut_pmc.sh <outdir-under-gpurun_out> [phase_cut.py args, e.g. --robot fetch --shelf]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cut in 7 8 1 2 3 0; do
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
              "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_WAVES"; do
    tag=c${cut}_$(echo $pass | cut -c1-12 | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/$tag -o p -- python $GRAFT_REPO_ROOT/tools/phase_cut.py --cut $cut "$@" > $out/$tag.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python - $out <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
GRID = int(os.environ.get("PHASE_CUT_GRID", int(os.environ.get("GTO_SLOTS", "512")) * 16 * 256))  # threads of the evaluation launch: B x 16 waypoint groups x 256
rows = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "c*_*"))):
    if not os.path.isdir(d):
        continue
    cut = os.path.basename(d).split("_")[0]
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_obstacle_gram" in r["Kernel_Name"] and int(r.get("Grid_Size", 0) or 0) == GRID:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[1:] if len(v) > 1 else v  # the first launch is cold
        rows[cut][k] = sum(v) / len(v)
    t = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if t:
        du = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(t[0]))
              if "k_obstacle_gram" in r["Kernel_Name"] and int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) == GRID]
        if du:
            rows[cut]["us"] = min(du) / 1e3
names = sorted({k for r in rows.values() for k in r})
order = ["c7", "c8", "c1", "c2", "c3", "c0"]
lines = ["cut (cumulative up to the end of: 7 table staging, 8 sin/cos, 1 kinematics, 2 broad phase, 3 gather loop, 0 whole kernel), per launch",
         f"{'counter':28s}" + "".join(f"{c:>14s}" for c in order)]
for n in names:
    lines.append(f"{n:28s}" + "".join(f"{rows[c].get(n, float('nan')):14.4g}" for c in order))
open(os.path.join(out, "phase_cut_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
for d in $out/c*_*; do [ -d "$d" ] && rm -rf $d; done
