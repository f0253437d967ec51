#!/bin/bash
# Copy the evidence set scripts/round_profiles.sh <tag> left under gpurun_out/ into profiles/ (tracked):
#   bash scripts/publish_profiles.sh <tag>
set -e
tag=$1
root=$(cd "$(dirname "$0")/.." && pwd)
g=$root/gpurun_out p=$root/profiles
cp $(ls -tr $g/${tag}_slots1/*/*kernel_stats.csv | tail -1) $p/${tag}_slots1_kernel_stats.csv
cp $(ls -tr $g/${tag}_default/*/*kernel_stats.csv | tail -1) $p/${tag}_default_bench_kernel_stats.csv
cp $g/${tag}_default_bench_line.json $p/${tag}_default_bench_line_under_rocprof.json
cp $g/${tag}_default_occupancy.txt $p/${tag}_default_bench_occupancy.txt
cp $g/${tag}_pmc_FETCH_SIZE.txt $g/${tag}_pmc_WRITE_SIZE.txt $p/
cat $g/${tag}_sq1.txt $g/${tag}_sq2.txt > $p/${tag}_pmc_SQ.txt
cp $g/${tag}_bench_line.json $p/${tag}_bench_line.json
cp $(ls -tr $g/${tag}_sgm/*/*kernel_stats.csv | tail -1) $p/${tag}_sgm_kernel_stats.csv
cp $g/${tag}_sgm_bench_line.json $p/${tag}_sgm_bench_line.json
cp $(ls -tr $g/${tag}_bm/*/*kernel_stats.csv | tail -1) $p/${tag}_bm_kernel_stats.csv
cp $g/${tag}_bm_bench_line.json $g/${tag}_bm_config2_bench_line.json $p/
[ -s $g/${tag}_other_configs.jsonl ] && cp $g/${tag}_other_configs.jsonl $p/
for f in qsad_probe node_rate host_pointer_rate latency_check lone_timeline sgm_dbg support_split_ab; do [ -s $g/${tag}_$f.txt ] && cp $g/${tag}_$f.txt $p/; done   # scripts/extra_evidence.sh
[ -s $g/${tag}_parity_sweep.txt ] && grep -v "Opened result" $g/${tag}_parity_sweep.txt > $p/r02_parity_sweep.txt
python3 $root/scripts/make_pmc_json.py $tag $p/r02_pmc_traffic.json
ls -la $p | grep ${tag}
