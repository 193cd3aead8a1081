#!/bin/bash
# Round evidence in one GPU call: bench lines (default, the driver's call), the rocprofv3 kernel-trace summaries of the same
# commands, the per-round listing of the driver's 20-step region, the latency tools and the PMC passes at both call sizes.
# Everything lands in gpurun_out/evidence/ with a stamp of the commit and of the library sources (lib_sha16) in every
# file name's companion STAMP and inside every text file; tools/run_evidence.sh (local) refuses to start from a dirty tree
# and copies what is worth judging to profiles/.
# usage (through tools/run_evidence.sh): tools/evidence.sh <tag> <git head>
tag=${1:-r06}; head=${2:-unknown}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/evidence; mkdir -p $out
cd $root
sha=$(python -c "import bench; print(bench.lib_sha16())")
stamp="# evidence $tag: commit $head, lib_sha16 $sha (sources of libgto_hip.so), $(date -u +%Y-%m-%dT%H:%MZ)"
echo "$stamp" > $out/${tag}_STAMP.txt
st() { for f in "$@"; do [ -f "$f" ] && sed -i "1i $stamp" "$f"; done; }
b() { name=$1; shift; timeout 900 python bench.py "$@" > $out/${tag}_$name.json 2> $out/${tag}_$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$out/${tag}_$name.json').read().strip().split('\n')[-1]); print(d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'gate', d.get('quality',{}).get('gate'), 'lib', d.get('lib_sha16'))
except Exception as e: print('unparsed', e)")"; }
b bench --gpus 1
b bench_steps20 --gpus 1 --steps 20 --warmup 5
cd /tmp && export TMPDIR=/tmp
prof() { name=$1; shift; rm -rf $out/prof_$name; timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_$name -o p -- python $root/bench.py "$@" --no-cpu-baseline > $out/${tag}_prof_$name.log 2>&1
  db=$(ls $out/prof_$name/*.db $out/prof_$name/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $root/tools/rocprof_summary.py $db $out/${tag}_kernel_stats_$name.md --title "bench.py $* (rocprofv3 --kernel-trace --stats)"; rm -rf $out/prof_$name; st $out/${tag}_kernel_stats_$name.md; }
prof one_lane --pipeline 1 --merged-launches-only
prof one_lane_steps20 --pipeline 1 --merged-launches-only --steps 20 --merge 5 --warmup 1
prof pipelined
prof pipelined_steps20 --steps 20 --warmup 5
rm -rf /tmp/tl_ev; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_ev -o p -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --merged-launches-only --repeats 2 > $out/${tag}_rounds.log 2>&1
csv=$(ls /tmp/tl_ev/*kernel_trace.csv /tmp/tl_ev/*/*kernel_trace.csv 2>/dev/null | head -1); python $root/tools/round_trace.py $csv 0 > $out/${tag}_rounds_steps20.txt; python $root/tools/timeline.py $csv 3 > $out/${tag}_timeline_steps20.txt 2>&1
# the same listing for BASELINE configs[2] and [4] (four lanes, calls of 8 steps: the lines of other_configs)
for cfgn in "cfg2 --robot fetch --batch 256 --shelf" "cfg4 --robot fetch_mobile --T 80 --grid 256 --shelf --batch 64"; do set -- $cfgn; nm=$1; shift
  rm -rf /tmp/tl_$nm; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$nm -o p -- python $root/bench.py --light --no-cpu-baseline --no-next-rows --no-other-configs --repeats 2 --warmup 1 --merge 8 --steps 32 "$@" > $out/${tag}_rounds_$nm.log 2>&1
  csv=$(ls /tmp/tl_$nm/*kernel_trace.csv /tmp/tl_$nm/*/*kernel_trace.csv 2>/dev/null | head -1); python $root/tools/round_trace.py $csv 0 | awk 'NR<=40 || NR%10==0' > $out/${tag}_rounds_$nm.txt; st $out/${tag}_rounds_$nm.txt; done
cd $root
B=1,4,8,64,150 python tools/serial_latency.py > $out/${tag}_serial_latency.txt 2>&1
python tools/planner_latency.py > $out/${tag}_planner_latency.txt 2>&1
python tools/pipeline_latency.py > $out/${tag}_pipeline_latency.txt 2>&1
st $out/${tag}_rounds_steps20.txt $out/${tag}_timeline_steps20.txt $out/${tag}_serial_latency.txt $out/${tag}_planner_latency.txt $out/${tag}_pipeline_latency.txt
# PMC passes (counters in runs of their own: --kernel-trace --pmc only) at the driver's call size and at the default's, and of
# BASELINE configs[2] (Fetch in the shelf) and configs[4] (mobile Fetch, 256^3) at the call sizes of the line's other_configs
bash tools/pmc_pass.sh pmc_320 20 5 > /dev/null 2>&1; cp gpurun_out/pmc_320/pmc_summary.txt $out/${tag}_pmc_320.txt
bash tools/pmc_pass.sh pmc_2048 32 32 > /dev/null 2>&1; cp gpurun_out/pmc_2048/pmc_summary.txt $out/${tag}_pmc_2048.txt
bash tools/pmc_pass.sh pmc_cfg2_2048 8 8 --robot fetch --batch 256 --shelf --light --no-next-rows --no-other-configs --repeats 1 > /dev/null 2>&1; cp gpurun_out/pmc_cfg2_2048/pmc_summary.txt $out/${tag}_pmc_cfg2_2048.txt
bash tools/pmc_pass.sh pmc_cfg4_512 8 8 --robot fetch_mobile --T 80 --grid 256 --shelf --batch 64 --light --no-next-rows --no-other-configs --repeats 1 > /dev/null 2>&1; cp gpurun_out/pmc_cfg4_512/pmc_summary.txt $out/${tag}_pmc_cfg4_512.txt
st $out/${tag}_pmc_320.txt $out/${tag}_pmc_2048.txt $out/${tag}_pmc_cfg2_2048.txt $out/${tag}_pmc_cfg4_512.txt
# stamped critical path of a step-kernel workgroup (GTO_DEBUG_TIMING), next to the PMC figures of the same variants
python tools/step_stamps.py $tag > $out/${tag}_step_stamps.log 2>&1; cp profiles/${tag}_step_stamps.json $out/ 2>/dev/null
python tools/traffic_json.py $tag gpurun_out/pmc_320 gpurun_out/pmc_2048 gpurun_out/pmc_cfg2_2048:fetch:128:shelf gpurun_out/pmc_cfg4_512:fetch_mobile:256:shelf > /dev/null && cp profiles/traffic.json $out/traffic.json
# where the obstacle kernel's instructions go on the shelf workload (cumulative cuts, PMC per cut)
bash tools/phase_cut_pmc.sh cut_cfg2 --robot fetch --shelf > /dev/null 2>&1; cp gpurun_out/cut_cfg2/phase_cut_pmc.txt $out/${tag}_phase_cut_cfg2.txt; st $out/${tag}_phase_cut_cfg2.txt
# where an iteration of the base-placement kernel goes (debug build with phase stamps)
python tools/base_stamps.py $out/${tag}_base_stamps.txt > /dev/null 2>&1; st $out/${tag}_base_stamps.txt
# what a workgroup of the obstacle kernel costs by its surviving chunks (debug build with per-workgroup clocks)
python tools/wg_model.py $out/${tag}_wg_model_cfg2.txt > /dev/null 2>&1; st $out/${tag}_wg_model_cfg2.txt
python tools/wg_model.py $out/${tag}_wg_model_default.txt --batch 512 > /dev/null 2>&1; st $out/${tag}_wg_model_default.txt
python tools/wg_model.py $out/${tag}_wg_model_cfg4.txt --robot fetch_mobile --T 80 --grid 256 --shelf --batch 64 --merge 8 > /dev/null 2>&1; st $out/${tag}_wg_model_cfg4.txt
# lanes inside one call (gto_set_lanes): kernel trace of one call of 1280 instances on four lanes, without and with the hand-over
cd /tmp
for ad in 0 48; do rm -rf /tmp/tl_ln; GTO_ADOPT=$ad timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_ln -o p -- python $root/bench.py --steps 20 --warmup 2 --repeats 2 --no-cpu-baseline --no-next-rows --merged-launches-only --pipeline 1 --merge 20 --lanes 4 > $out/${tag}_lanes.log 2>&1
  csv=$(ls /tmp/tl_ln/*kernel_trace.csv /tmp/tl_ln/*/*kernel_trace.csv 2>/dev/null | head -1); python $root/tools/lane_trace.py $csv 1280 > $out/${tag}_lanes_one_call_adopt$ad.txt 2>&1; st $out/${tag}_lanes_one_call_adopt$ad.txt; done
cd $root
# the rerun of the line with the fresh traffic.json / stamps in place (roofline.variants carry the PMC figures of THIS commit)
b bench_steps20 --gpus 1 --steps 20 --warmup 5
b bench --gpus 1
ls -la $out
