#!/bin/bash
# SQ-side PMC passes of one bench.py run (inside gpurun): instruction mix and wait buckets per kernel.
#   bash scripts/pmc_sq.sh <tag> <kernel regex> [bench args...]      (env vars pass through)
# Two passes of <= 8 SQ counters each (MI355X_MICROARCH.md, rocprofv3 PMC slots); --pmc is never combined with
# tracing domains other than --kernel-trace.
tag=$1; shift
pat=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
p1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
p2="SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
i=0
for set in "$p1" "$p2"; do
  i=$((i+1))
  python3 $root/scripts/fresh_dir.py gpurun_out/${tag}_sq$i; timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $root/gpurun_out/${tag}_sq$i -- python3 $root/bench.py --steps 3 --warmup 1 --slots 1 --no-cpu-baseline --no-latency-config --no-alone-leg --min-time 0 "$@" > $root/gpurun_out/${tag}_sq$i.log 2>&1
  python3 $root/scripts/pmc.py $(ls $root/gpurun_out/${tag}_sq$i/*/*counter_collection.csv | tail -1) "$pat" > $root/gpurun_out/${tag}_sq$i.txt
  cat $root/gpurun_out/${tag}_sq$i.txt
done
