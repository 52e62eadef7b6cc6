#!/bin/bash
# Local: copies the files tools/refresh_evidence.sh (and tools/prof_all.sh) produced from gpurun_out/ (scratch) into profiles/
# (tracked) and regenerates profiles/SUMMARY.md.
set -e
TAG=${1:-r3}
cd "$(dirname "$0")/.."
for f in pmc_step_kernel_hand.json pmc_step_kernel_hand_B32768.json pmc_step_kernel_legs.json pmc_step_kernel_trackenv.json kernel_stats_hand.csv kernel_stats_bench.csv \
         kernel_stats_hand_B32768.csv kernel_stats_legs.csv kernel_stats_trackenv.csv bench_line_under_rocprof.json \
         stage_stamps_hand.txt stage_stamps_legs.txt stage_stamps_trackenv.txt bench_line.json bench_line_B32768.json bench_line_legs.json \
         bench_line_config2_posefixed_1000steps.json bench_line_reach.json bench_line_terrain.json bench_line_objhold.json \
         bench_line_finger.json bench_line_trackenv.json; do
  [ -s gpurun_out/${TAG}_$f ] && cp gpurun_out/${TAG}_$f profiles/${TAG}_$f || echo "missing gpurun_out/${TAG}_$f"
done
[ -f gpurun_out/parity_report.json ] && cp gpurun_out/parity_report.json profiles/${TAG}_parity_report.json
python tools/make_summary.py $TAG > /dev/null
echo collected
