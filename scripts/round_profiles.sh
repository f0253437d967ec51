#!/bin/bash
# The round's evidence set in one gpurun call:  bash scripts/round_profiles.sh <tag>
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
bash scripts/collect_profiles.sh $tag > gpurun_out/${tag}_collect.log 2>&1
bash scripts/pmc_sq.sh $tag "k_dense|k_support_lds|k_descriptor" > gpurun_out/${tag}_sq.log 2>&1
# SGM mode: kernel stats + bench line
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_sgm -- python3 $root/bench.py --mode sgm --steps 5 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${tag}_sgm.log 2>&1
cd $root
python3 scripts/kstats.py $(ls gpurun_out/${tag}_sgm/*/*kernel_stats.csv | head -1) 8 > gpurun_out/${tag}_sgm_summary.txt
python3 bench.py --mode sgm --steps 10 --warmup 2 > gpurun_out/${tag}_sgm_bench_line.json 2> gpurun_out/${tag}_sgm_bench.err
tail -12 gpurun_out/${tag}_collect.log | cut -c1-400; cat gpurun_out/${tag}_sgm_summary.txt; cut -c1-300 gpurun_out/${tag}_sgm_bench_line.json
