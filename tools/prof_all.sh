# rocprofv3 evidence of the round (GPU box): per workload a kernel trace (--kernel-trace --stats) and PMC passes, every counter group in its own
# pass (never combined with trace domains), all through the small driver tools/prof_step.py (the program sits directly after `--`); the
# headline additionally gets the kernel trace of bench.py itself.
# usage (from the repo root on the box):  bash tools/prof_all.sh r3 [what ...]     what = hand hand32k legs trackenv (default: all)
#   -> gpurun_out/r3_pmc_step_kernel_<what>.json, r3_kernel_stats_<what>.csv  (copy what is to be judged into profiles/: tools/collect_evidence.sh)
set -e
TAG=${1:-r3}; shift || true
WHAT=${@:-hand hand32k legs trackenv}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
prof() {  # suffix env B kernel-filter algorithmic-bytes-per-launch [full]
  sfx=$1; export ENV=$2; export B=$3; kf=$4; alg=$5
  D=$O/prof_${TAG}_$sfx; rm -rf ${D}_*
  rocprofv3 --kernel-trace --stats --output-format csv -d ${D}_trace -- python3 $R/tools/prof_step.py > ${D}_trace.log 2>&1
  pmc() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d ${D}_$n -- python3 $R/tools/prof_step.py > ${D}_$n.log 2>&1 || echo "pmc pass $sfx/$n failed (see ${D}_$n.log)"; }
  pmc fetch FETCH_SIZE
  pmc write WRITE_SIZE
  pmc pmc1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
  pmc pmc2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
  pmc pmc3 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
  pmc pmc4 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY
  pmc pmc5 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT
  if [ "$6" = "full" ]; then
    pmc pmc6 GRBM_GUI_ACTIVE SQ_INSTS_FLAT SQ_WAVES_EQ_64
    pmc pmc7 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum
    pmc pmc8 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
    pmc pmc9 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SMEM
  fi
  (cd $R && SKIP=30 python3 tools/prof_collect.py $O/${TAG}_pmc_step_kernel_$sfx.json "$kf" $alg ${D}_trace ${D}_fetch ${D}_write ${D}_pmc1 ${D}_pmc2 ${D}_pmc3 ${D}_pmc4 ${D}_pmc5 $( [ "$6" = "full" ] && echo ${D}_pmc6 ${D}_pmc7 ${D}_pmc8 ${D}_pmc9 ))
  cp $(find ${D}_trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_$sfx.csv
  echo "profiled $sfx"
}
for w in $WHAT; do
  case $w in
    hand)
      rocprofv3 -L > $O/${TAG}_counters_available.txt 2>&1 || true
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_benchtrace -- python3 $R/bench.py --steps 100 --warmup 30 --no-cpu-baseline > $O/prof_${TAG}_benchtrace.log 2>&1
      grep "^{\"metric\"" $O/prof_${TAG}_benchtrace.log | tail -1 > $O/${TAG}_bench_line_under_rocprof.json
      cp $(find $O/prof_${TAG}_benchtrace -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_bench.csv
      prof hand myoHandPoseRandom-v0 4096 "step_kernel_w<24" 6389760 full ;;
    hand32k) prof hand_B32768 myoHandPoseRandom-v0 32768 "step_kernel_w<24" 51118080 ;;
    legs) prof legs myoLegWalk-v0 4096 "step_kernel_w<36" 13975552 ;;
    trackenv) prof trackenv MyoHandAirplaneRandom-v0 4096 "step_kernel_w<36" 6864896 ;;
  esac
done
