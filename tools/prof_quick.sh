# quick PMC refresh of the headline step kernel (GPU box): instruction mix and wait shares;  bash tools/prof_quick.sh TAG [ENV] [B]
TAG=${1:-q}; export ENV=${2:-myoHandPoseRandom-v0}; export B=${3:-4096}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $O/pq_${TAG}_*
pmc() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/pq_${TAG}_$n -- python3 $R/tools/prof_step.py > $O/pq_${TAG}_$n.log 2>&1 || echo "pmc pass $n failed"; }
pmc pmc1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pmc pmc2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pmc pmc3 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
pmc pmc4 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY
pmc pmc5 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT
cd $R
SKIP=30 python3 tools/prof_collect.py $O/${TAG}_pmc_quick.json "step_kernel_w<" 0 $O/pq_${TAG}_pmc1 $O/pq_${TAG}_pmc2 $O/pq_${TAG}_pmc3 $O/pq_${TAG}_pmc4 $O/pq_${TAG}_pmc5
