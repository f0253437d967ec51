#!/bin/bash
# A slot's life in the pipelined path: rocprofv3 kernel trace of short bench runs (GPU route, host route, six slots) through scripts/slot_timeline.py.
R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in "1 4" "0 4" "1 6"; do
  set -- $cfg
  python3 $R/scripts/fresh_dir.py $R/gpurun_out/tl_$1_$2
  JN_GPU_DELAUNAY=$1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$1_$2 -- python3 $R/bench.py --steps 60 --warmup 8 --min-time 0.2 --slots $2 --no-cpu-baseline --no-latency-config --no-alone-leg > $R/gpurun_out/tl_$1_$2.log 2>&1
  echo "=== JN_GPU_DELAUNAY=$1 slots $2: $(grep -o '"value": [0-9.]*' $R/gpurun_out/tl_$1_$2.log | head -1)"
  python3 $R/scripts/slot_timeline.py $(ls $R/gpurun_out/tl_$1_$2/*/*kernel_trace.csv | tail -1)
  rm -rf $R/gpurun_out/tl_$1_$2
done
