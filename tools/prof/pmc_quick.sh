# instruction counters of the raster kernels for one library build: bash tools/prof/pmc_quick.sh <tag> [lib.so]  -> gpurun_out/<tag>_pmcq.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=$1
[ -n "$2" ] && export LASR_HIP_LIB=$R/$2
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0 --steps 3 --warmup 1"
rm -f $O/${T}_pmcq.txt
pmc() {
  rocprofv3 --pmc "$@" -d $O/pmc_x -o p -- $B > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find $O/pmc_x -name "*.db" | head -1) 2>/dev/null | grep -v "setup\|^#" >> $O/${T}_pmcq.txt; rm -rf $O/pmc_x
}
pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD
pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU_INT32
pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32
sort $O/${T}_pmcq.txt | cut -c1-110
