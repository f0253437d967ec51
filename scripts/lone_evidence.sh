#!/bin/bash
# Lone-pair evidence (inside gpurun; part of scripts/round4_profiles.sh):  bash scripts/lone_evidence.sh <tag>
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
out=gpurun_out
fresh() { python3 $R/scripts/fresh_dir.py gpurun_out/$1; }
# lone pair: stage B queued behind the gate (default for latency-mode handles) against queued after the host stage; then one pair's kernel timeline each way
{ for g in 1 0 1 0; do echo "JN_GATE_STAGE_B=$g: $(JN_GATE_STAGE_B=$g HT=8 LONE_ONLY=1 timeout 200 python3 scripts/latency_check.py 2>/dev/null | grep "device pointers" | cut -c1-110)"; done; } > $out/${tag}_gate_ab.txt
cd /tmp
for g in 1 0; do
  fresh ${tag}_lone_g$g; JN_GATE_STAGE_B=$g HT=8 LONE_ONLY=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$out/${tag}_lone_g$g -- python3 $R/scripts/latency_check.py > /dev/null 2>&1
done
cd $R
{ for g in 1 0; do echo "== JN_GATE_STAGE_B=$g (under rocprofv3 --kernel-trace: the host side is slower than in a plain run, the gaps are what to read) =="; python3 scripts/lone_timeline.py $(ls $out/${tag}_lone_g$g/*/*kernel_trace.csv | tail -1); done; } > $out/${tag}_lone_timeline.txt 2>&1
