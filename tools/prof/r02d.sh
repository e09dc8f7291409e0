R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/r02d_pytest.txt
python bench.py > $O/r02d_bench.json 2> $O/r02d_bench.err
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_$c -o p -- python $R/bench.py --no-cpu-baseline --no-lbs --lasr-iters 0 --steps 5 --warmup 1 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find $O/pmc_$c -name "*.db" | head -1) >> $O/r02d_pmc.txt; rm -rf $O/pmc_$c
done
cat $O/r02d_pytest.txt; python -c "
import json;d=json.load(open('$O/r02d_bench.json'))
print(d['value'], d['roofline']['all_kernels_avg_ms'], d['roofline']['frac']); print(d['cpu_baseline']); print(d['lbs']); print(d.get('optimize_py',{}).get('iters_per_s'))"
cat $O/r02d_pmc.txt | cut -c1-130
