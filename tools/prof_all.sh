set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/prof_r1f_* $O/prof_r1g_*
# hand (headline workload): kernel trace of bench.py itself, then counters on the same workload via the torch-free driver
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r1f_trace -- python3 $R/bench.py --steps 50 --warmup 30 --no-cpu-baseline > $O/prof_r1f_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_r1f_fetch -- python3 $R/tools/prof_step.py > $O/prof_r1f_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_r1f_write -- python3 $R/tools/prof_step.py > $O/prof_r1f_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/prof_r1f_pmc1 -- python3 $R/tools/prof_step.py > $O/prof_r1f_pmc1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/prof_r1f_pmc2 -- python3 $R/tools/prof_step.py > $O/prof_r1f_pmc2.log 2>&1
cd $R
SKIP=30 python3 tools/prof_collect.py $O/r1_f_pmc_step_kernel_hand.json "step_kernel_w<24" 6389760 $O/prof_r1f_trace $O/prof_r1f_fetch $O/prof_r1f_write $O/prof_r1f_pmc1 $O/prof_r1f_pmc2
cp $(find $O/prof_r1f_trace -name "*kernel_stats.csv" | head -1) $O/r1_f_kernel_stats_hand.csv
tail -1 $O/prof_r1f_trace.log > $O/r1_f_bench_line_under_rocprof.json
# legs
cd /tmp
ENV=myoLegWalk-v0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r1g_trace -- python3 $R/tools/prof_step.py > $O/prof_r1g_trace.log 2>&1
ENV=myoLegWalk-v0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_r1g_fetch -- python3 $R/tools/prof_step.py > $O/prof_r1g_fetch.log 2>&1
ENV=myoLegWalk-v0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_r1g_write -- python3 $R/tools/prof_step.py > $O/prof_r1g_write.log 2>&1
cd $R
SKIP=30 python3 tools/prof_collect.py $O/r1_g_pmc_step_kernel_legs.json "step_kernel_w<36" 13975552 $O/prof_r1g_trace $O/prof_r1g_fetch $O/prof_r1g_write
cp $(find $O/prof_r1g_trace -name "*kernel_stats.csv" | head -1) $O/r1_g_kernel_stats_legs.csv
