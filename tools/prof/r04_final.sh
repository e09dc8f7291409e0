# the measurement set of the round: bash tools/prof/r04_final.sh <tag>   (one MI355X)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04}
cd $R
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0"
rocprofv3 --kernel-trace -d $O/prof_o -o o -- python $R/bench.py --no-cpu-baseline --no-lbs --no-sweep --steps 1 --warmup 1 --frames 16 --lasr-iters 30 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_o -name "*.db" | head -1) 90 > $O/${T}_optimize_step_kernel_stats.txt
python $R/tools/step_sequence.py $(find $O/prof_o -name "*.db" | head -1) > $O/${T}_step_sequence.txt; rm -rf $O/prof_o
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
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $O/pmc_x -o p -- python $R/tools/lbs_bench.py > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find $O/pmc_x -name "*.db" | head -1) > $O/${T}_pmc_lbs.txt 2>/dev/null; rm -rf $O/pmc_x
python $R/tools/valu_json.py $O/${T}_pmc_sq.txt 256 > $O/${T}_valu.json
# the launch sizes the reference uses: kernel trace + occupancy counters at 16 and 4 frames per launch
for n in 16 4; do
  rocprofv3 --kernel-trace -d $O/prof_k -o k -- $B --frames $n --steps 200 --warmup 5 > /dev/null 2>&1
  python $R/tools/rocpd_stats.py $(find $O/prof_k -name "*.db" | head -1) > $O/${T}_kernel_stats_n$n.txt; rm -rf $O/prof_k
  rm -f $O/${T}_pmc_sq_n$n.txt
  pmc $O/${T}_pmc_sq_n$n.txt "--frames $n" SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE
done
cd $R; python tools/op_census.py 2>&1 | grep -v -i warn > $O/${T}_op_census.txt
python tools/traffic_json.py $O/${T}_pmc.txt 256 "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py --no-cpu-baseline --no-lbs --lasr-iters 0 --steps 3 --warmup 1 (256 frames per launch, mesh M2, 256x256); KiB per dispatch summed over all TCC instances; bytes = FETCH_SIZE x 1024 x read_factor + WRITE_SIZE x 1024 x write_factor with the factors of profiles/r04_traffic_calibration.json (tools/ubench/traffic.hip: vector reads are tallied at half their bytes, scalar-cache reads and stores in full)" $R/profiles/r04_traffic_calibration.json > $O/${T}_traffic.json
cp $O/${T}_traffic.json $O/${T}_valu.json $R/profiles/
# The bench lines and the kernel trace they must agree with are NOT taken here: a box that has just run the GPU test-suite and
# these profiling passes measured every kernel ~6-8 % slower (62.4 k instead of 66 k frames/s; clocks, not code).  Copy
# gpurun_out/${T}_* into profiles/ and run tools/prof/r04_bench.sh ${T} in a SEPARATE gpurun call (fresh box): bench.py then
# quotes the counter files of this build, and the rocprofv3 kernel trace of the same command is taken right after it.
cat $O/${T}_pmc.txt | cut -c1-120; cat $O/${T}_pmc_lbs.txt | grep -i lbs | cut -c1-130
