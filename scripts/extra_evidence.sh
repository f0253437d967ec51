#!/bin/bash
# Small text evidence for numbers DESIGN.md quotes outside bench.py (inside gpurun):  bash scripts/extra_evidence.sh <tag>
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/probes/qsad_probe.hip -o /tmp/qsad_probe && timeout 60 /tmp/qsad_probe > $out/${tag}_qsad_probe.txt 2>&1
timeout 300 python3 scripts/node_rate.py 300 > $out/${tag}_node_rate.txt 2>&1
timeout 400 python3 scripts/host_pointer_rate.py > $out/${tag}_host_pointer_rate.txt 2>&1
HT=8 timeout 200 python3 scripts/latency_check.py > $out/${tag}_latency_check.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && LONE_ONLY=1 HT=8 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $root/$out/${tag}_lone -- python3 $root/scripts/latency_check.py > /dev/null 2>&1 )
python3 scripts/lone_timeline.py $(ls $out/${tag}_lone/*/*kernel_trace.csv | head -1) > $out/${tag}_lone_timeline.txt 2>&1
for d in 0 1; do JN_SGM_DBG=$d python3 bench.py --mode sgm --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('JN_SGM_DBG=$d', j['stage_ms_per_batch'])"; done > $out/${tag}_sgm_dbg.txt 2>&1
for rep in 1 2; do for v in 0 1; do JN_SUPPORT_SPLIT=$v python3 bench.py --no-cpu-baseline --no-latency-config 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('JN_SUPPORT_SPLIT=$v (0 = default segments, 1 = one workgroup per lattice row)', j['value'], 'pairs/s', j['ms_per_step'], 'ms/step, gpu_support', j['stage_ms_per_batch']['gpu_support'], 'ms')"; done; done > $out/${tag}_support_split_ab.txt 2>&1
timeout 900 python3 scripts/parity_sweep.py 12 > $out/${tag}_parity_sweep.txt 2>&1
tail -3 $out/${tag}_parity_sweep.txt; cat $out/${tag}_qsad_probe.txt $out/${tag}_node_rate.txt $out/${tag}_sgm_dbg.txt $out/${tag}_support_split_ab.txt; tail -4 $out/${tag}_lone_timeline.txt
