#!/bin/bash
# Runs on the MI355X box (gpurun): regenerates every evidence file of profiles/ into gpurun_out/.
# Afterwards, locally: bash tools/collect_evidence.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/prof_all.sh
python tools/gpu_stamps.py > gpurun_out/r1_f_stage_stamps_hand.txt 2>&1
MYO_SCHED=0 ENV=myoLegWalk-v0 python tools/gpu_stamps.py > gpurun_out/r1_g_stage_stamps_legs.txt 2>&1
python bench.py > gpurun_out/r1_f_bench_line.json 2> gpurun_out/bench_err.log
python bench.py --batch 32768 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r1_f_bench_line_B32768.json
python bench.py --env myoLegWalk-v0 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r1_g_bench_line_legs.json
python bench.py --env myoHandPoseFixed-v0 --steps 1000 --warmup 50 --repeats 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r1_f_bench_line_config2_posefixed_1000steps.json
python bench.py --env myoHandReachRandom-v0 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r1_f_bench_line_reach.json
python bench.py --env myoLegRoughTerrainWalk-v0 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r1_h_bench_line_terrain.json
python bench.py --env myoHandObjHoldFixed-v0 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r1_h_bench_line_objhold.json
python bench.py --env myoFingerPoseFixed-v0 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r1_h_bench_line_finger.json
python tools/parity_report.py > gpurun_out/parity_report.log 2>&1 || true
grep -h '^{"metric"' gpurun_out/prof_r1f_trace.log > gpurun_out/r1_f_bench_line_under_rocprof.json
echo evidence refreshed
