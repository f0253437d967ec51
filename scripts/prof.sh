#!/bin/bash
# Profile one bench.py run on the GPU box with rocprofv3 (kernel trace + stats) and print the per-kernel
# summary.  Usage (inside gpurun):  bash scripts/prof.sh <tag> [bench args...]   (env vars pass through)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python3 $root/scripts/fresh_dir.py gpurun_out/$tag; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag -- python3 $root/bench.py --steps 10 --warmup 2 --slots 1 --no-cpu-baseline --no-latency-config "$@" > $root/gpurun_out/$tag.log 2>&1
python3 $root/scripts/kstats.py $(ls $root/gpurun_out/$tag/*/*kernel_stats.csv | tail -1) 24
