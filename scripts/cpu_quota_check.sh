#!/bin/bash
# How much of the container's CPU quota (cgroup v2 cpu.max) a bench.py run uses and whether it was throttled (inside gpurun):
#   bash scripts/cpu_quota_check.sh [bench args...]
cg=/sys/fs/cgroup
echo "cpu.max: $(cat $cg/cpu.max 2>/dev/null)   loadavg: $(cat /proc/loadavg)"
a=$(cat $cg/cpu.stat); t0=$(date +%s.%N)
python3 bench.py --no-cpu-baseline --no-latency-config --no-alone-leg "$@" > gpurun_out/quota.json 2>/dev/null
t1=$(date +%s.%N); b=$(cat $cg/cpu.stat)
python3 - "$a" "$b" "$t0" "$t1" <<'P'
import sys, json
def parse(s): return {l.split()[0]: int(l.split()[1]) for l in s.strip().split("\n")}
a, b = parse(sys.argv[1]), parse(sys.argv[2]); wall = float(sys.argv[4]) - float(sys.argv[3])
j = json.load(open("gpurun_out/quota.json"))
print("bench %.0f pairs/s, %.3f ms/step, host_stage %.2f ms | whole process: wall %.1f s, cpu %.1f s = %.1f cores on average; periods %d, throttled %d (%.2f s)" %
      (j["value"], j["ms_per_step"], j["stage_ms_per_batch"]["host_stage"], wall, (b["usage_usec"] - a["usage_usec"]) / 1e6,
       (b["usage_usec"] - a["usage_usec"]) / 1e6 / wall, b["nr_periods"] - a["nr_periods"], b["nr_throttled"] - a["nr_throttled"],
       (b["throttled_usec"] - a["throttled_usec"]) / 1e6))
P
