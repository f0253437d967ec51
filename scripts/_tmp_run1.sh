timeout 900 python -m pytest tests/test_gpu_sgm.py -x -q -m gpu 2>&1 | tail -3
run() { python3 bench.py --mode sgm --sgm-slots $1 --steps 18 --warmup 6 --no-cpu-baseline $2 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms per batch", j["check"]["ok"], j.get("stage_ms_per_batch"))'; }
for ss in 5 6 7 6; do echo "$ss slots: $(run $ss)"; done
echo "6 slots, JN_SGM_OVERLAP=1: $(JN_SGM_OVERLAP=1 run 6)"
echo "4 slots, JN_SGM_OVERLAP=1: $(JN_SGM_OVERLAP=1 run 4)"
echo "1080p D=256 subpixel batch 8, 4 slots: $(run 4 '--width 1920 --height 1080 --disp 256 --batch 8 --subpixel 1')"
echo "1080p D=256 subpixel batch 8, 6 slots: $(run 6 '--width 1920 --height 1080 --disp 256 --batch 8 --subpixel 1')"
echo "1080p D=256 subpixel batch 8, 1 slots: $(run 1 '--width 1920 --height 1080 --disp 256 --batch 8 --subpixel 1')"
