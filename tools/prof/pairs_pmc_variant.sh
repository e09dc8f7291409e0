# VALU / SALU / LDS instruction counts of the pair-walk forward kernel in a measurement build:
#   bash tools/prof/pairs_pmc_variant.sh <variant .so name under lasr_amd/csrc/variants | shipped> [frames]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; V=$1; N=${2:-256}
cd /tmp; export TMPDIR=/tmp
[ "$V" != shipped ] && export LASR_HIP_LIB=$R/lasr_amd/csrc/variants/$V
LASR_SR_PAIR_MIN_TILES=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU -d $O/pmc_v -o p -- python $R/tools/prof/pairs_check.py time-child $N 3 > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find $O/pmc_v -name "*.db" | head -1) 2>/dev/null | grep "forward" | awk -v v=$V '{print v, $2, $3, $5}'; rm -rf $O/pmc_v
