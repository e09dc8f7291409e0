# the measurement set of the round: bash tools/prof/r06_final.sh <tag>   (one MI355X; counters + step traces; the bench lines come
# from tools/prof/r06_bench.sh on a FRESH box afterwards -- a box that has just run these passes clocks 6-8 % lower)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=${1:-r06}
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0"
# ---- one optimisation iteration, per kernel (the trace bench.py's in_scope_step block is built from) and in launch order
for cfg in spot3_s0 camel_s4; do
  rocprofv3 --kernel-trace -d $O/prof_o -o o -- python $R/bench.py --step-worker $cfg > /dev/null 2>&1
  db=$(find $O/prof_o -name "*.db" | head -1)
  if [ $cfg = spot3_s0 ]; then
    python $R/tools/rocpd_stats.py $db 60 > $O/${T}_optimize_step_kernel_stats.txt
    python $R/tools/step_sequence.py $db > $O/${T}_step_sequence.txt
  else
    python $R/tools/rocpd_stats.py $db 60 > $O/${T}_optimize_step_kernel_stats_camel_s4.txt
  fi
  rm -rf $O/prof_o
done
pmc() {  # $1 = output file, $2 = extra bench args, rest = counters
  out=$1; extra=$2; shift; shift
  rocprofv3 --pmc "$@" -d $O/pmc_x -o p -- $B $extra --steps 3 --warmup 1 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find $O/pmc_x -name "*.db" | head -1) >> $out 2>/dev/null; rm -rf $O/pmc_x
}
rm -f $O/${T}_pmc.txt $O/${T}_pmc_sq.txt $O/${T}_pmc_512.txt $O/${T}_pmc_lbs.txt
pmc $O/${T}_pmc.txt "" FETCH_SIZE
pmc $O/${T}_pmc.txt "" WRITE_SIZE
pmc $O/${T}_pmc_512.txt "--image-size 512 --frames 64" FETCH_SIZE
pmc $O/${T}_pmc_512.txt "--image-size 512 --frames 64" WRITE_SIZE
pmc $O/${T}_pmc_sq.txt "" SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD
pmc $O/${T}_pmc_sq.txt "" SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_BRANCH
pmc $O/${T}_pmc_sq.txt "" SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32
# ---- LBS: MFMA counters of the forward AND the backward (tools/lbs_bench.py runs both)
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $O/pmc_x -o p -- python $R/tools/lbs_bench.py > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find $O/pmc_x -name "*.db" | head -1) > $O/${T}_pmc_lbs.txt 2>/dev/null; rm -rf $O/pmc_x
python $R/tools/valu_json.py $O/${T}_pmc_sq.txt 256 > $O/${T}_valu.json
cd $R
python tools/traffic_json.py $O/${T}_pmc.txt 256 "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0 --steps 3 --warmup 1 (256 frames per launch, mesh M2, 256x256); KiB per dispatch summed over all TCC instances; bytes = FETCH_SIZE x 1024 x read_factor + WRITE_SIZE x 1024 x write_factor with the factors of profiles/r04_traffic_calibration.json (tools/ubench/traffic.hip: vector reads are tallied at half their bytes, scalar-cache reads and stores in full)" $R/profiles/r04_traffic_calibration.json > $O/${T}_traffic.json
# ---- issue- or latency-bound?  the one-wave forward kernel at 8 / 6 / 4 / 2 waves per SIMD
# (needs the occ6 / occ4 / occ2 builds of tools/prof/occupancy_sweep.sh under lasr_amd/csrc/variants/; skipped when they are absent)
[ -f lasr_amd/csrc/variants/liblasr_hip_occ4.so ] && bash tools/prof/occupancy_sweep.sh > $O/${T}_occupancy_sweep.txt 2>&1
cat $O/${T}_pmc.txt | cut -c1-120; grep -i lbs $O/${T}_pmc_lbs.txt | cut -c1-130; [ -f $O/${T}_occupancy_sweep.txt ] && grep -v Traceback $O/${T}_occupancy_sweep.txt | head -20
