#!/bin/bash
# PMC counters of the dominant kernel in separate rocprofv3 passes (<= 8 SQ, <= 4 TCC counters per pass; FETCH_SIZE and
# WRITE_SIZE cannot share one), on bench.py with one lane and only calls of the timed region's size.
# usage: tools/pmc_pass.sh <outdir-under-gpurun_out> <steps> <merge> [more bench args]
#   driver's config (calls of 320 instances):   tools/pmc_pass.sh pmc_320 20 5
#   default config  (calls of 2048 instances):  tools/pmc_pass.sh pmc_2048 32 32
out=$GRAFT_REPO_ROOT/gpurun_out/$1; steps=$2; merge=$3; shift 3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o p -- python $GRAFT_REPO_ROOT/bench.py --pipeline 1 --steps $steps --merge $merge --warmup 1 --no-cpu-baseline --merged-launches-only $EXTRA > $out/$name.log 2>&1; }
EXTRA="$*"
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out
python tools/rocprof_summary.py $out/sq1 > $out/kernel_stats.md 2>/dev/null || true
# the raw per-dispatch CSVs are tens of MiB: only the summaries travel back from the GPU box (gpurun_out is capped at 64 MiB)
if [ -z "$GTO_PMC_KEEP_RAW" ]; then for d in sq1 sq2 tcc fetch write grbm; do rm -rf $out/$d; done; fi
