R=$GRAFT_REPO_ROOT; cd $R
bash tools/prof/ab_many.sh lasr_amd/csrc/liblasr_hip.so lasr_amd/csrc/variants/liblasr_hip_ilv2.so lasr_amd/csrc/variants/liblasr_hip_ilv4.so lasr_amd/csrc/variants/liblasr_hip_ilv8.so lasr_amd/csrc/variants/liblasr_hip_ilv16.so 2>&1 | head -5
cd /tmp; export TMPDIR=/tmp
for l in liblasr_hip.so variants/liblasr_hip_ilv4.so variants/liblasr_hip_ilv8.so; do
 LASR_HIP_LIB=$R/lasr_amd/csrc/$l rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_x -o p -- python $R/bench.py --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0 --steps 3 --warmup 1 > /dev/null 2>&1
 echo $l; python $R/tools/pmc_summary.py $(find $R/gpurun_out/pmc_x -name "*.db" | head -1) | grep "forward.*ELb0ELb0ELb1" ; rm -rf $R/gpurun_out/pmc_x
done
