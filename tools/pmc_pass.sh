#!/bin/bash
# Collect PMC counters for the dominant kernel in separate passes (rocprofv3 --pmc; <=8 SQ, <=4 TCC per pass).
# usage: tools/pmc_pass.sh <outdir-under-gpurun_out> [bench args...]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o p -- python $GRAFT_REPO_ROOT/bench.py --pipeline 1 --steps 32 --warmup 1 --no-cpu-baseline --merged-launches-only $BENCH_ARGS > $out/$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out
