#!/bin/bash
# Local: copies the files tools/refresh_evidence.sh produced from gpurun_out/ (scratch) into profiles/ (tracked) and regenerates
# profiles/SUMMARY.md.
set -e
cd "$(dirname "$0")/.."
for f in r1_f_pmc_step_kernel_hand.json r1_f_kernel_stats_hand.csv r1_f_stage_stamps_hand.txt r1_f_bench_line.json r1_f_bench_line_B32768.json \
         r1_f_bench_line_config2_posefixed_1000steps.json r1_f_bench_line_reach.json r1_f_bench_line_under_rocprof.json \
         r1_g_pmc_step_kernel_legs.json r1_g_kernel_stats_legs.csv r1_g_stage_stamps_legs.txt r1_g_bench_line_legs.json \
         r1_h_bench_line_terrain.json r1_h_bench_line_objhold.json r1_h_bench_line_finger.json; do
  cp gpurun_out/$f profiles/$f
done
[ -f gpurun_out/parity_report.json ] && cp gpurun_out/parity_report.json profiles/r1_parity_report.json
python tools/make_summary.py > /dev/null
echo collected
