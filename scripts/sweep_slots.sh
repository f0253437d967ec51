for cfg in "4 16" "4 24" "4 32" "6 24" "6 32" "8 32" "4 48"; do set -- $cfg
python bench.py --no-cpu-baseline --no-latency-config --no-alone-leg --slots $1 --host-threads $2 > gpurun_out/sw.json 2>/dev/null
python -c "
import json;j=json.load(open('gpurun_out/sw.json'));print('slots $1 threads $2', j['value'], j['ms_per_step'], j['stage_ms_per_batch']['host_stage'], j['stage_ms_per_batch']['total'])"
done
