#!/bin/bash
# The round's evidence set in one gpurun call:  bash scripts/round_profiles.sh <tag>
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
bash scripts/collect_profiles.sh $tag > gpurun_out/${tag}_collect.log 2>&1
bash scripts/pmc_sq.sh $tag "k_dense|k_support_lds|k_descriptor" > gpurun_out/${tag}_sq.log 2>&1
# SGM mode: kernel stats + bench line
cd /tmp && export TMPDIR=/tmp
python3 $root/scripts/fresh_dir.py gpurun_out/${tag}_sgm; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_sgm -- python3 $root/bench.py --mode sgm --steps 5 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${tag}_sgm.log 2>&1
cd $root
python3 scripts/kstats.py $(ls gpurun_out/${tag}_sgm/*/*kernel_stats.csv | tail -1) 8 > gpurun_out/${tag}_sgm_summary.txt
python3 bench.py --mode sgm --steps 10 --warmup 2 > gpurun_out/${tag}_sgm_bench_line.json 2> gpurun_out/${tag}_sgm_bench.err
# block-matching mode: kernel stats + bench lines (BASELINE config 2 = lone 640x480 D=64, and the headline workload's shape)
cd /tmp
python3 $root/scripts/fresh_dir.py gpurun_out/${tag}_bm; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_bm -- python3 $root/bench.py --mode bm --steps 10 --warmup 2 --no-cpu-baseline > $root/gpurun_out/${tag}_bm.log 2>&1
cd $root
python3 scripts/kstats.py $(ls gpurun_out/${tag}_bm/*/*kernel_stats.csv | tail -1) 8 > gpurun_out/${tag}_bm_summary.txt
python3 bench.py --mode bm --steps 20 --warmup 3 > gpurun_out/${tag}_bm_bench_line.json 2> gpurun_out/${tag}_bm_bench.err
python3 bench.py --mode bm --width 640 --height 480 --disp 64 --batch 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${tag}_bm_config2_bench_line.json 2>> gpurun_out/${tag}_bm_bench.err
# the north star's other frame sizes (pairs/s, pipelined, device-resident inputs), one JSON line each
: > gpurun_out/${tag}_other_configs.jsonl
for a in "--width 640 --height 480 --disp 64 --batch 32" "--width 640 --height 480 --disp 64 --batch 64" "--width 320 --height 180 --disp 256 --scene-disp 48 --batch 128" "--width 1920 --height 1080 --disp 256 --batch 8"; do
  python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-latency-config $a >> gpurun_out/${tag}_other_configs.jsonl 2>> gpurun_out/${tag}_bench.err
done
for m in sgm bm; do python3 bench.py --mode $m --width 640 --height 480 --disp 64 --batch 32 --steps 20 --warmup 3 --no-cpu-baseline >> gpurun_out/${tag}_other_configs.jsonl 2>> gpurun_out/${tag}_bench.err; done
tail -12 gpurun_out/${tag}_collect.log | cut -c1-400; cat gpurun_out/${tag}_sgm_summary.txt; cut -c1-300 gpurun_out/${tag}_sgm_bench_line.json; cat gpurun_out/${tag}_bm_summary.txt; cut -c1-300 gpurun_out/${tag}_bm_bench_line.json gpurun_out/${tag}_bm_config2_bench_line.json
