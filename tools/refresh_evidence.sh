#!/bin/bash
# Runs on the MI355X box (gpurun): regenerates every evidence file of profiles/ into gpurun_out/ under the round tag (default r2).
# Afterwards, locally: bash tools/collect_evidence.sh r2
# PROF=0 skips the rocprofv3 passes (tools/prof_all.sh), e.g. when they were already run in their own gpurun call.
set -e
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
[ "${PROF:-1}" = "1" ] && bash tools/prof_all.sh $TAG
python tools/gpu_stamps.py > $O/${TAG}_stage_stamps_hand.txt 2>&1
MYO_SCHED=0 ENV=myoLegWalk-v0 python tools/gpu_stamps.py > $O/${TAG}_stage_stamps_legs.txt 2>&1
python tools/gpu_track_stamps.py > $O/${TAG}_stage_stamps_trackenv.txt 2>&1 || true
python bench.py > $O/${TAG}_bench_line.json 2> $O/bench_err.log
b() { out=$1; shift; python bench.py "$@" --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${TAG}_bench_line_$out.json; echo "$out done"; }
b B32768 --batch 32768 --steps 120 --warmup 10
b legs --env myoLegWalk-v0
b config2_posefixed_1000steps --env myoHandPoseFixed-v0 --steps 1000 --warmup 50 --repeats 5
b reach --env myoHandReachRandom-v0
b terrain --env myoLegRoughTerrainWalk-v0
b objhold --env myoHandObjHoldFixed-v0
b finger --env myoFingerPoseFixed-v0
b trackenv --env MyoHandAirplaneRandom-v0 --steps 200 --warmup 20
python tools/parity_report.py > $O/parity_report.log 2>&1 || true
echo evidence refreshed
