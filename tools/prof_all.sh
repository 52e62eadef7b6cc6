# rocprofv3 evidence of the round (GPU box): kernel trace of bench.py itself, then PMC passes on the same workload through the torch-free
# driver tools/prof_step.py (the program sits directly after `--`; counters in their own passes, never combined with trace domains).
# usage (from the repo root on the box):  bash tools/prof_all.sh r2      -> gpurun_out/r2_* (copy what is to be judged into profiles/)
set -e
TAG=${1:-r2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/prof_${TAG}_*
rocprofv3 -L > $O/${TAG}_counters_available.txt 2>&1 || true
# hand (headline workload)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_trace -- python3 $R/bench.py --steps 100 --warmup 30 --no-cpu-baseline > $O/prof_${TAG}_trace.log 2>&1
pmc() {  # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $O/prof_${TAG}_$n -- python3 $R/tools/prof_step.py > $O/prof_${TAG}_$n.log 2>&1 || echo "pmc pass $n failed (see prof_${TAG}_$n.log)"
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc pmc1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pmc pmc2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pmc pmc3 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
pmc pmc4 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY
pmc pmc5 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT
pmc pmc6 GRBM_GUI_ACTIVE SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_I8 SQ_WAVES_EQ_64
pmc pmc7 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum
pmc pmc8 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
pmc pmc9 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SMEM
cd $R
SKIP=30 python3 tools/prof_collect.py $O/${TAG}_pmc_step_kernel_hand.json "step_kernel_w<24" 6389760 $O/prof_${TAG}_trace $O/prof_${TAG}_fetch $O/prof_${TAG}_write $O/prof_${TAG}_pmc1 $O/prof_${TAG}_pmc2 $O/prof_${TAG}_pmc3 $O/prof_${TAG}_pmc4 $O/prof_${TAG}_pmc5 $O/prof_${TAG}_pmc6 $O/prof_${TAG}_pmc7 $O/prof_${TAG}_pmc8 $O/prof_${TAG}_pmc9
cp $(find $O/prof_${TAG}_trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_hand.csv
grep "^{\"metric\"" $O/prof_${TAG}_trace.log | tail -1 > $O/${TAG}_bench_line_under_rocprof.json
# legs
cd /tmp
ENV=myoLegWalk-v0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}g_trace -- python3 $R/tools/prof_step.py > $O/prof_${TAG}g_trace.log 2>&1
ENV=myoLegWalk-v0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_${TAG}g_fetch -- python3 $R/tools/prof_step.py > $O/prof_${TAG}g_fetch.log 2>&1
ENV=myoLegWalk-v0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_${TAG}g_write -- python3 $R/tools/prof_step.py > $O/prof_${TAG}g_write.log 2>&1
ENV=myoLegWalk-v0 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --output-format csv -d $O/prof_${TAG}g_pmc3 -- python3 $R/tools/prof_step.py > $O/prof_${TAG}g_pmc3.log 2>&1 || true
cd $R
SKIP=30 python3 tools/prof_collect.py $O/${TAG}_pmc_step_kernel_legs.json "step_kernel_w<36" 13975552 $O/prof_${TAG}g_trace $O/prof_${TAG}g_fetch $O/prof_${TAG}g_write $O/prof_${TAG}g_pmc3
cp $(find $O/prof_${TAG}g_trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_legs.csv
