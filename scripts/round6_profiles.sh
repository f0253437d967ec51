#!/bin/bash
# The round's evidence set in ONE gpurun call:  bash scripts/round6_profiles.sh <tag>      (e.g. r04; ONE tag per round: a refresh overwrites the same files)
# Every rocprofv3 output directory is emptied (scripts/fresh_dir.py) before the run that fills it; nothing here combines --pmc
# with a tracing domain other than --kernel-trace.  scripts/publish_round.py <tag> then copies the set into profiles/.
tag=$1
[ -n "$tag" ] || { echo "usage: round6_profiles.sh <tag>"; exit 1; }
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
HOOKS=$R/jackal_navigation_amd/libjn_stereo_hooks.so    # the A/B knobs and profiling switches live in the hooks build only (csrc/hooks.h): lines that use one load it
out=gpurun_out
fresh() { python3 $R/scripts/fresh_dir.py gpurun_out/$1; }
JN_PMC_ALL=1 bash scripts/collect_profiles.sh $tag > $out/${tag}_collect.log 2>&1
bash scripts/pmc_sq.sh $tag "k_dense_row|k_support_lds|k_sobel|k_owner|k_gap_mean|k_scan<|k_ccl|k_lr_ccl|k_bin|k_delaunay" > $out/${tag}_sq.log 2>&1
# ---- SGM mode: kernel stats, bench line, PMC traffic (FETCH / WRITE in separate passes), SQ pass ----
cd /tmp && export TMPDIR=/tmp
fresh ${tag}_sgm; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/${tag}_sgm -- python3 $R/bench.py --mode sgm --sgm-slots 1 --steps 5 --warmup 1 --no-cpu-baseline > $R/$out/${tag}_sgm.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  fresh ${tag}_sgm_pmc_$c; timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$out/${tag}_sgm_pmc_$c -- python3 $R/bench.py --mode sgm --sgm-slots 1 --steps 2 --warmup 1 --no-cpu-baseline > $R/$out/${tag}_sgm_pmc_$c.log 2>&1
done
fresh ${tag}_sgm_sq; timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $R/$out/${tag}_sgm_sq -- python3 $R/bench.py --mode sgm --sgm-slots 1 --steps 2 --warmup 1 --no-cpu-baseline > $R/$out/${tag}_sgm_sq.log 2>&1
cd $R
python3 scripts/kstats.py $(ls $out/${tag}_sgm/*/*kernel_stats.csv | tail -1) 8 > $out/${tag}_sgm_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do python3 scripts/pmc.py $(ls $out/${tag}_sgm_pmc_$c/*/*counter_collection.csv | tail -1) "k_sw_" > $out/${tag}_sgm_pmc_$c.txt; done
python3 scripts/pmc.py $(ls $out/${tag}_sgm_sq/*/*counter_collection.csv | tail -1) "k_sw_" > $out/${tag}_sgm_pmc_SQ.txt
python3 bench.py --mode sgm --steps 60 --warmup 6 > $out/${tag}_sgm_bench_line.json 2> $out/${tag}_sgm_bench.err
{ for ss in 1 2 4 6 8 6 1; do echo "--sgm-slots $ss: $(python3 bench.py --mode sgm --sgm-slots $ss --steps 12 --warmup 6 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms per batch, frac", j["roofline"]["frac"])')"; done
  echo "--sgm-slots 6, JN_SGM_TAIL=3 (L/R kernel, mono8 kernel, scan one after the other): $(JN_SGM_TAIL=3 python3 bench.py --mode sgm --sgm-slots 6 --steps 12 --warmup 6 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms per batch")')"
  for ss in 1 4 6; do echo "1920x1080 D=256 + 1/16 px batch 8, --sgm-slots $ss: $(python3 bench.py --mode sgm --sgm-slots $ss --width 1920 --height 1080 --disp 256 --batch 8 --subpixel 1 --steps 8 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms per batch")')"; done
  for lq in 8 4; do echo "1920x1080 D=256 + 1/16 px batch 8, one batch at a time, JN_SGM_LQ=$lq: $(JN_STEREO_LIB=$HOOKS JN_SGM_LQ=$lq python3 bench.py --mode sgm --sgm-slots 1 --width 1920 --height 1080 --disp 256 --batch 8 --subpixel 1 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s", j["stage_ms_per_batch"])')"; done; } > $out/${tag}_sgm_slots_ab.txt
{ for ns in 2 4 8; do echo "JN_SGM_NS=$ns $(JN_STEREO_LIB=$HOOKS JN_SGM_NS=$ns python3 bench.py --mode sgm --sgm-slots 1 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s", j["stage_ms_per_batch"])')"; done; } > $out/${tag}_sgm_strips_ab.txt
# ---- block-matching mode ----
cd /tmp
fresh ${tag}_bm; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/${tag}_bm -- python3 $R/bench.py --mode bm --bm-slots 1 --steps 10 --warmup 2 --no-cpu-baseline > $R/$out/${tag}_bm.log 2>&1
cd $R
python3 scripts/kstats.py $(ls $out/${tag}_bm/*/*kernel_stats.csv | tail -1) 8 > $out/${tag}_bm_summary.txt
python3 bench.py --mode bm --steps 20 --warmup 3 > $out/${tag}_bm_bench_line.json 2> $out/${tag}_bm_bench.err
python3 bench.py --mode bm --bm-slots 1 --width 640 --height 480 --disp 64 --batch 1 --steps 200 --warmup 20 --no-cpu-baseline > $out/${tag}_bm_config2_bench_line.json 2>> $out/${tag}_bm_bench.err
# ---- block matching with the squared-difference cost on the matrix cores (csrc/bm_mfma.hip), next to the v_qsad kernel ----
cd /tmp
fresh ${tag}_bm_ssd; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/${tag}_bm_ssd -- python3 $R/bench.py --mode bm --bm-cost ssd --bm-slots 1 --steps 10 --warmup 2 --no-cpu-baseline > $R/$out/${tag}_bm_ssd.log 2>&1
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $R/$out/${tag}_mfma_counters_available.txt
fresh ${tag}_bm_ssd_pmc; timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA --output-format csv -d $R/$out/${tag}_bm_ssd_pmc -- python3 $R/bench.py --mode bm --bm-cost ssd --bm-slots 1 --steps 2 --warmup 1 --no-cpu-baseline > $R/$out/${tag}_bm_ssd_pmc.log 2>&1
cd $R
python3 scripts/kstats.py $(ls $out/${tag}_bm_ssd/*/*kernel_stats.csv | tail -1) 8 > $out/${tag}_bm_ssd_summary.txt
{ echo "# counters with MFMA in their name on this box: $(tr '\n' ' ' < $out/${tag}_mfma_counters_available.txt)"; python3 scripts/pmc.py $(ls $out/${tag}_bm_ssd_pmc/*/*counter_collection.csv | tail -1) "k_bmq_"; } > $out/${tag}_bm_ssd_pmc_mfma.txt 2>&1
python3 bench.py --mode bm --bm-cost ssd --steps 20 --warmup 3 > $out/${tag}_bm_ssd_bench_line.json 2>> $out/${tag}_bm_bench.err
{ for c in sad ssd; do for ss in 1 2 4 6 1 4; do echo "--mode bm --bm-cost $c --bm-slots $ss: $(python3 bench.py --mode bm --bm-cost $c --bm-slots $ss --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms per batch")')"; done; done; } > $out/${tag}_bm_slots_ab.txt
python3 bench.py --mode bm --bm-cost ssd --width 1920 --height 1080 --disp 256 --batch 8 --steps 20 --warmup 3 --no-cpu-baseline > $out/${tag}_bm_ssd_1080p_bench_line.json 2>> $out/${tag}_bm_bench.err
python3 bench.py --mode bm --bm-cost sad --width 1920 --height 1080 --disp 256 --batch 8 --steps 20 --warmup 3 --no-cpu-baseline > $out/${tag}_bm_sad_1080p_bench_line.json 2>> $out/${tag}_bm_bench.err
# ---- the north star's other frame sizes, one JSON line each ----
: > $out/${tag}_other_configs.jsonl
for a in "--width 640 --height 480 --disp 64 --batch 32" "--width 640 --height 480 --disp 64 --batch 64" "--width 320 --height 180 --disp 256 --scene-disp 48 --batch 128" "--width 1920 --height 1080 --disp 256 --batch 8"; do
  python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-latency-config $a 2>> $out/${tag}_bench.err | grep '^{"metric"' >> $out/${tag}_other_configs.jsonl
done
for m in sgm bm; do python3 bench.py --mode $m --width 640 --height 480 --disp 64 --batch 32 --steps 24 --warmup 6 --no-cpu-baseline 2>> $out/${tag}_bench.err | grep '^{"metric"' >> $out/${tag}_other_configs.jsonl; done
python3 bench.py --mode sgm --width 1920 --height 1080 --disp 256 --batch 8 --subpixel 1 --steps 12 --warmup 4 --no-cpu-baseline 2>> $out/${tag}_bench.err | grep '^{"metric"' >> $out/${tag}_other_configs.jsonl
# ---- merge in the slot worker (one-rank communicator), node rate, lone-pair latency, host-pointer rates, probes ----
for i in 1 2; do python3 bench.py --no-cpu-baseline --force-merge --no-latency-config 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); m=j["merge"]; print(j["value"], "pairs/s with the merge;", m["merge_ms_per_step"], "ms per step;", m["pairs_per_sec_without_merge"], "pairs/s without; cost", m["cost_frac"])'; done > $out/${tag}_merge_in_worker.txt 2>&1
timeout 300 python3 scripts/node_rate.py 300 serial 2>/dev/null | tail -1 > $out/${tag}_node_rate.txt; timeout 300 python3 scripts/node_rate.py 300 2>/dev/null | tail -1 >> $out/${tag}_node_rate.txt
HT=8 timeout 200 python3 scripts/latency_check.py 2>/dev/null | grep -v amdgpu.ids > $out/${tag}_latency_check.txt
timeout 400 python3 scripts/host_pointer_rate.py 2>/dev/null | grep -v amdgpu.ids > $out/${tag}_host_pointer_rate.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/probes/valu_rate_probe.hip -o /tmp/valu_rate_probe && timeout 60 /tmp/valu_rate_probe > $out/${tag}_valu_rate_probe.txt 2>&1
for pr in pk3 dep_chain lds_unaligned op_rate lone_wave lds_misaligned_store; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/probes/${pr}_probe.hip -o /tmp/${pr}_probe && timeout 120 /tmp/${pr}_probe > $out/${tag}_${pr}_probe.txt 2>&1; done
# k_dense_row's phases by its JN_DENSE_DBG switches (hooks build; results wrong, timing only): the kernel alone, ms per launch by HIP events
{ for d in 0 1 2 3 4 12 20 28 64 0; do echo "JN_DENSE_DBG=$d: $(JN_STEREO_LIB=$HOOKS JN_DENSE_DBG=$d python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-latency-config 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(r.get("ms_per_launch"), "ms alone,", r.get("ms_per_launch_pipelined"), "ms pipelined,", j["value"], "pairs/s")')"; done; } > $out/${tag}_dense_dbg_switches.txt
bash scripts/hwq_ab.sh > $out/${tag}_hw_queues_ab.txt 2>&1
bash scripts/lone_evidence.sh $tag
bash scripts/host_threads_sweep.sh "2 4 8 12 16" > $out/${tag}_host_threads.txt 2>&1
for v in 0 1; do echo "JN_STAGE_A_PRIORITY=$v $(JN_STAGE_A_PRIORITY=$v python3 bench.py --no-cpu-baseline --no-latency-config 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s", j["ms_per_step"], "ms/step")')"; done > $out/${tag}_stage_a_priority_ab.txt
for v in 1 0 1 0; do echo "JN_PACE=$v, the driver's command (--gpus 1 --steps 20 --warmup 5): $(JN_PACE=$v python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency-config 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s", j["ms_per_step"], "ms/step")')"; done > $out/${tag}_pace_ab.txt
for v in 1 0; do echo "JN_PACE=$v, default 200 steps per region: $(JN_PACE=$v python3 bench.py --no-cpu-baseline --no-latency-config 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s", j["ms_per_step"], "ms/step")')"; done >> $out/${tag}_pace_ab.txt
timeout 1200 python3 scripts/parity_sweep.py 12 2>&1 | grep -v "Opened result\|amdgpu.ids" > $out/${tag}_parity_sweep.txt
timeout 400 python3 scripts/sgm_stress.py 120 2>&1 | grep -v amdgpu.ids > $out/${tag}_sgm_stress.txt
cat $out/${tag}_bm_ssd_summary.txt $out/${tag}_bm_ssd_pmc_mfma.txt | head -30; tail -8 $out/${tag}_collect.log | cut -c1-300; cat $out/${tag}_sgm_summary.txt $out/${tag}_sgm_strips_ab.txt $out/${tag}_merge_in_worker.txt $out/${tag}_node_rate.txt; tail -2 $out/${tag}_parity_sweep.txt
# ---- round 6: FETCH_SIZE against known bytes, the non-volatile k_delaunay build, a slot's life in the pipeline, 1920x1080 on both routes ----
bash scripts/probes/fetch_size_probe.sh > $out/${tag}_fetch_size_probe.txt 2>&1
bash scripts/probes/dt_no_volatile.sh > $out/${tag}_dt_no_volatile.txt 2>&1
bash scripts/slot_timeline.sh > $out/${tag}_slot_timeline.txt 2>&1
bash scripts/full_hd_routes.sh > $out/${tag}_full_hd_routes.txt 2>&1
# ---- round 5: triangulation on the GPU against the host stage by core count, k_delaunay's levels, the lone pair under a few settings ----
bash scripts/gpu_delaunay_ab.sh > $out/${tag}_gpu_delaunay_ab.txt 2>&1
bash scripts/lone_env_ab.sh > $out/${tag}_lone_env_ab.txt 2>&1
python3 -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $out/${tag}_gpu_tests.txt; cat $out/${tag}_gpu_tests.txt
