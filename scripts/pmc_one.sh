#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the kernels matching a regex, one-slot bench (inside gpurun): bash scripts/pmc_one.sh <regex>
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  python3 $root/scripts/fresh_dir.py gpurun_out/pmc_one_$c; timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $root/gpurun_out/pmc_one_$c -- python3 $root/bench.py --steps 3 --warmup 1 --slots 1 --no-cpu-baseline --no-latency-config > /dev/null 2>&1
  python3 $root/scripts/pmc.py $(ls $root/gpurun_out/pmc_one_$c/*/*counter_collection.csv | tail -1) "$1"
done
