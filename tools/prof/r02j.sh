R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python tools/cosdist_bench.py > $O/r02_cosdist.json 2>/dev/null; cat $O/r02_cosdist.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof_o -o o -- python $R/bench.py --no-cpu-baseline --no-lbs --steps 1 --warmup 1 --frames 16 --lasr-iters 30 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_o -name "*.db" | head -1) 90 > $O/r02j_optimize_step_kernel_stats.txt; rm -rf $O/prof_o
head -30 $O/r02j_optimize_step_kernel_stats.txt | cut -c1-150
