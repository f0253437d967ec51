# A/B of an environment switch inside one gpurun call:  bash scripts/ab.sh VAR a b [reps]
var=$1; a=$2; b=$3; reps=${4:-3}
for i in $(seq $reps); do for v in $a $b; do
env $var=$v python bench.py --no-cpu-baseline --no-latency-config > gpurun_out/ab.json 2>/dev/null
python -c "
import json;j=json.load(open('gpurun_out/ab.json'));print('$var=$v', j['value'], j['ms_per_step'], j['roofline']['ms_per_launch'], j['check']['ok'])"
done; done
