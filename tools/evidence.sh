#!/bin/bash
# Round evidence in one GPU call: bench lines (default, the driver's call, the other BASELINE configurations), the rocprofv3
# kernel-trace summaries of the same commands and the per-round listing of the driver's 20-step region.  (The single-launch
# mode is a test-only second implementation since round 3: no bench line, no profile.)  Everything lands in
# gpurun_out/evidence/; the files worth judging are copied to profiles/ by hand (named per round).
# usage: tools/evidence.sh [tag]      (run through gpurun from the repo root)
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/evidence; mkdir -p $out
cd $root
b() { name=$1; shift; timeout 600 python bench.py "$@" > $out/${tag}_$name.json 2> $out/${tag}_$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$out/${tag}_$name.json').read().strip().split('\n')[-1]); print(d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'gate', d.get('quality',{}).get('gate'))
except Exception as e: print('unparsed', e)")"; }
b bench --gpus 1
b bench_steps20 --gpus 1 --steps 20 --warmup 5
b cfg2_fetch_shelf --gpus 1 --robot fetch --batch 256 --shelf --merge 8 --cpu-seconds 8
b cfg4_fetch_mobile --gpus 1 --robot fetch_mobile --T 80 --grid 256 --shelf --batch 64 --merge 8 --steps 32 --cpu-seconds 8
b cfg3_scene_sharded --gpus 1 --scene-sharded --scenes-per-gpu 256 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
prof() { name=$1; shift; rm -rf $out/prof_$name; timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_$name -o p -- python $root/bench.py "$@" --no-cpu-baseline > $out/${tag}_prof_$name.log 2>&1
  db=$(ls $out/prof_$name/*.db $out/prof_$name/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $root/tools/rocprof_summary.py $db $out/${tag}_kernel_stats_$name.md --title "bench.py $* (rocprofv3 --kernel-trace --stats)"; rm -rf $out/prof_$name; }
prof one_lane --pipeline 1 --merged-launches-only
prof one_lane_steps20 --pipeline 1 --merged-launches-only --steps 20 --merge 5 --warmup 1
prof pipelined
prof pipelined_steps20 --steps 20 --warmup 5
rm -rf /tmp/tl_ev; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_ev -o p -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --merged-launches-only --repeats 2 > $out/${tag}_rounds.log 2>&1
csv=$(ls /tmp/tl_ev/*kernel_trace.csv | head -1); python $root/tools/round_trace.py $csv 0 > $out/${tag}_rounds_steps20.txt; python $root/tools/timeline.py $csv 3 > $out/${tag}_timeline_steps20.txt 2>&1
python $root/tools/serial_latency.py > $out/${tag}_serial_latency.txt 2>&1
python $root/tools/planner_latency.py > $out/${tag}_planner_latency.txt 2>&1
python $root/tools/pipeline_latency.py > $out/${tag}_pipeline_latency.txt 2>&1
ls -la $out
