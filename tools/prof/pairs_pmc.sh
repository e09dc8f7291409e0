# counters of the forward kernels, pair-walk vs one wave per tile: bash tools/prof/pairs_pmc.sh <tag> [frames]  -> gpurun_out/<tag>_pairs_pmc.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=$1; N=${2:-256}
cd /tmp; export TMPDIR=/tmp
rm -f $O/${T}_pairs_pmc.txt
pmc() {
  for mode in 0 1000000000000; do
    LASR_SR_PAIR_MIN_TILES=$mode rocprofv3 --pmc "$@" -d $O/pmc_x -o p -- python $R/tools/prof/pairs_check.py time-child $N 3 > /dev/null 2>&1
    python $R/tools/pmc_summary.py $(find $O/pmc_x -name "*.db" | head -1) 2>/dev/null | grep "forward" >> $O/${T}_pairs_pmc.txt; rm -rf $O/pmc_x
  done
}
pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD
pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU_INT32
pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32
pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT64 SQ_LEVEL_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
sort $O/${T}_pairs_pmc.txt | cut -c1-120
