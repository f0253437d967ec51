#!/bin/bash
# Hardware queues and the pipelined SGM mode (inside gpurun).  A HIP stream is served by one of GPU_MAX_HW_QUEUES hardware queues; streams
# beyond that share queues, and two SGM slots on one queue run their batches one after the other.
#  (1) slots made to share streams on purpose (JN_SGM_STREAMS=k: slot s queues on the stream of slot s % k);
#  (2) the SGM handle next to an OPEN four-slot ELAS handle (what bench.py's other_modes leg did before it closed that handle first).
for cfg in "0 4" "3 6" "2 4" "1 4"; do
  set -- $cfg
  JN_SGM_STREAMS=$1 python3 bench.py --mode sgm --sgm-slots $2 --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('--mode sgm --sgm-slots $2, JN_SGM_STREAMS=$1 (0 = one stream per slot):', d['value'], 'pairs/s')"
done
for q in 8 16; do
  GPU_MAX_HW_QUEUES=$q python3 scripts/sgm_next_to_elas.py 2>/dev/null | tail -1 | sed "s/^/GPU_MAX_HW_QUEUES=$q: /"
done
