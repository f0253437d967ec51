#!/bin/bash
# Collect the round's rocprofv3 evidence in one go (run inside gpurun from the repo root):
#   bash scripts/collect_profiles.sh <tag>        e.g. r01_final
# Writes under gpurun_out/<tag>_*: kernel stats for one slot (kernels back to back) and for the default pipelined
# bench, the two PMC passes for HBM traffic (FETCH_SIZE and WRITE_SIZE in separate runs, as the MI355X guide
# prescribes), the pipeline occupancy and the bench JSON line.
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $root/scripts/fresh_dir.py gpurun_out/${tag}_slots1; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_slots1 -- python3 $root/bench.py --steps 10 --warmup 2 --slots 1 --no-cpu-baseline --no-latency-config > $out/${tag}_slots1.log 2>&1
python3 $root/scripts/fresh_dir.py gpurun_out/${tag}_default; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_default -- python3 $root/bench.py --no-cpu-baseline --no-latency-config > $out/${tag}_default.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  python3 $root/scripts/fresh_dir.py gpurun_out/${tag}_pmc_$c; timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${tag}_pmc_$c -- python3 $root/bench.py --steps 3 --warmup 1 --slots 1 --no-cpu-baseline --no-latency-config > $out/${tag}_pmc_$c.log 2>&1
done
cd $root
python3 scripts/kstats.py $(ls $out/${tag}_slots1/*/*kernel_stats.csv | tail -1) 30 > $out/${tag}_slots1_summary.txt
python3 scripts/kstats.py $(ls $out/${tag}_default/*/*kernel_stats.csv | tail -1) 30 > $out/${tag}_default_summary.txt
python3 scripts/busy.py $(ls $out/${tag}_default/*/*kernel_trace.csv | tail -1) > $out/${tag}_default_occupancy.txt
# JN_PMC_ALL=1 (round 5 on): every kernel of the path, not only the heavy four
pat="k_dense|k_descriptor|k_support|k_lr$"; [ -n "$JN_PMC_ALL" ] && pat="^(void )?k_"
for c in FETCH_SIZE WRITE_SIZE; do python3 scripts/pmc.py $(ls $out/${tag}_pmc_$c/*/*counter_collection.csv | tail -1) "$pat" > $out/${tag}_pmc_$c.txt; done
grep "^{\"metric\"" $out/${tag}_default.log | tail -1 > $out/${tag}_default_bench_line.json
python3 bench.py > $out/${tag}_bench_line.json 2> $out/${tag}_bench.err
cat $out/${tag}_slots1_summary.txt | head -8; cat $out/${tag}_default_occupancy.txt | head -3; cat $out/${tag}_pmc_FETCH_SIZE.txt $out/${tag}_pmc_WRITE_SIZE.txt; tail -1 $out/${tag}_bench_line.json | cut -c1-300
