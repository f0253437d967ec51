#!/bin/bash
# One GPU round for the ELAS path:  bash scripts/elas_round.sh <tag> [quick]
#   parity tests, default bench line (+ an A/B of the stage-A stream priority), kernel stats with one slot
tag=${1:-elas}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
if [ "$2" != "quick" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.txt 2>&1; tail -3 gpurun_out/${tag}_pytest.txt; fi
for i in 1 2; do python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("default", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["check"]["ok"])'; done
JN_STAGE_A_PRIORITY=0 python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("no priority", j["value"], j["ms_per_step"])'
bash scripts/prof.sh ${tag}_slots1 | head -16
