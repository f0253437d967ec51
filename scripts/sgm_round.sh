#!/bin/bash
# One GPU round for the SGM mode: staged check, pytest, bench line, rocprof kernel stats.  bash scripts/sgm_round.sh <tag>
tag=${1:-sgm}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python scripts/sgm_debug.py > gpurun_out/${tag}_debug.txt 2>&1; echo "debug rc=$?"; grep -E "differ|ALL OK|FAIL|Error|error" gpurun_out/${tag}_debug.txt | head -20
timeout 600 python -m pytest tests/test_gpu_sgm.py -x -q > gpurun_out/${tag}_pytest.txt 2>&1; tail -2 gpurun_out/${tag}_pytest.txt
timeout 300 python bench.py --mode sgm --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench.txt 2>&1; tail -1 gpurun_out/${tag}_bench.txt | cut -c1-420
python3 $R/scripts/fresh_dir.py gpurun_out/${tag}_prof; cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python3 $R/bench.py --mode sgm --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_prof_bench.txt 2>&1; echo prof rc=$?
python3 $R/scripts/kstats.py $(ls $R/gpurun_out/${tag}_prof/*/*kernel_stats.csv | head -1) 6
