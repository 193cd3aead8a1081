#!/bin/bash
# Local side of the round's evidence: refuses a dirty tree (every number must belong to a commit), runs tools/evidence.sh on
# the GPU box through gpurun and copies the judged files into profiles/ (named per round, each carrying the stamp).
# usage: tools/run_evidence.sh [tag]
tag=${1:-r06}
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain --untracked-files=no)" ]; then echo "tree is dirty: commit first (evidence is stamped with the commit)"; git status --short; exit 1; fi
head=$(git rev-parse --short=12 HEAD)
python -c "import __graft_entry__ as g; g.build()" || exit 1
/usr/local/graft/bin/gpurun --timeout 3000 -- "bash tools/evidence.sh $tag $head > gpurun_out/evidence_$tag.log 2>&1; tail -25 gpurun_out/evidence_$tag.log" || exit 1
ev=gpurun_out/evidence
for f in $ev/${tag}_bench.json $ev/${tag}_bench_steps20.json $ev/${tag}_kernel_stats_*.md $ev/${tag}_rounds_steps20.txt $ev/${tag}_timeline_steps20.txt \
         $ev/${tag}_serial_latency.txt $ev/${tag}_planner_latency.txt $ev/${tag}_pipeline_latency.txt $ev/${tag}_pmc_320.txt $ev/${tag}_pmc_2048.txt $ev/${tag}_pmc_cfg2_2048.txt \
         $ev/${tag}_pmc_cfg4_512.txt $ev/${tag}_step_stamps.json $ev/${tag}_base_stamps.txt $ev/${tag}_wg_model_cfg2.txt $ev/${tag}_wg_model_default.txt $ev/${tag}_wg_model_cfg4.txt $ev/${tag}_phase_cut_cfg2.txt $ev/${tag}_lanes_one_call_adopt0.txt $ev/${tag}_lanes_one_call_adopt48.txt $ev/${tag}_rounds_cfg2.txt $ev/${tag}_rounds_cfg4.txt $ev/${tag}_STAMP.txt; do
  [ -f "$f" ] && cp "$f" profiles/
done
[ -f $ev/traffic.json ] && cp $ev/traffic.json profiles/traffic.json
ls profiles | grep "^${tag}_"
