#!/bin/bash
# pairs/s of the headline workload against the host cores a rank is given (VERDICT r03 #4c): the host stage (Delaunay's hull recursion,
# one task per frame and side) is the one part of the path that does not shard with the GPUs, so a node with fewer than 16 CPUs per GPU
# runs below the single-GPU figure.  Each line: `taskset -c 0-(T-1) python bench.py --host-threads T --no-pin` (slot workers and the
# submitting thread share the same T cores).    bash scripts/host_threads_sweep.sh "4 8 12 16" > gpurun_out/<tag>_host_threads.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "# cores visible to the box: $(nproc), cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for t in ${1:-"4 8 12 16"}; do
  line=$(taskset -c 0-$((t-1)) python bench.py --host-threads $t --no-pin --steps 20 --warmup 5 --no-cpu-baseline --no-latency-config --no-alone-leg 2>/dev/null | tail -1)
  echo "host threads $t on cores 0-$((t-1)): $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("%.0f pairs/s, %.3f ms per step, host cores busy %s" % (j["value"], j["ms_per_step"], j.get("host_cpu", {}).get("cores_total")))')"
done
