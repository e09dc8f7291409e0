# usage: bash tools/prof/pmc_fwd.sh <tag> <forward-variant>     -> gpurun_out/<tag>_pmc.txt (+ <tag>_bench.json)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; TAG=$1; V=$2
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --forward-variant $V --no-cpu-baseline --no-lbs --lasr-iters 0 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
pmc() {
  tag=$1; shift
  rocprofv3 --pmc "$@" -d $O/pmc_$tag -o p -- python $R/bench.py --forward-variant $V --no-cpu-baseline --no-lbs --lasr-iters 0 --steps 2 --warmup 1 > /dev/null 2>$O/pmc_$tag.err
  python $R/tools/pmc_summary.py $(find $O/pmc_$tag -name "*.db" | head -1) 2>>$O/pmc_$tag.err | grep -v setup >> $O/${TAG}_pmc.txt; rm -rf $O/pmc_$tag
}
rm -f $O/${TAG}_pmc.txt
pmc a SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD
pmc b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD
pmc c SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32
pmc d TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS
pmc e TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ
pmc f SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_IFETCH SQ_INSTS_VALU_INT64
python -c "import json;d=json.load(open('$O/${TAG}_bench.json'));print('$TAG', d['value'], d['roofline']['all_kernels_avg_ms'])"
grep -v backward $O/${TAG}_pmc.txt | cut -c1-130
