#!/bin/bash
# kernel trace of a bench.py run + tools/timeline.py on it.  usage (through gpurun): tools/timeline.sh <name> [bench args]
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/timeline; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$name
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$name -o p -- python $root/bench.py --no-cpu-baseline --merged-launches-only "$@" > $out/$name.log 2>&1
csv=$(ls /tmp/tl_$name/*kernel_trace.csv /tmp/tl_$name/*/*kernel_trace.csv 2>/dev/null | head -1)
python $root/tools/timeline.py $csv 3 | tee $out/$name.txt
