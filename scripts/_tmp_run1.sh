timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/pytest_tmp.txt 2>&1
echo "rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest_tmp.txt | tail -3
for i in 1 2; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], j["check"]["ok"] if "check" in j else None, j["latency_config"]["ms_per_frame"])'; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof1; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --slots 1 --steps 10 --warmup 2 --no-cpu-baseline --no-latency-config > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 scripts/kstats.py $(ls /tmp/prof1/*/*kernel_stats.csv | tail -1) 16
