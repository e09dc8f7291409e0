R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_raster_parity_gpu.py -m gpu -x -q 2>&1 | tail -8 > $O/r02b_pytest.txt
timeout 600 python -m pytest tests/test_lasr_forward_oracle_gpu.py -m gpu -q 2>&1 | tail -30 >> $O/r02b_pytest.txt
for v in 0 1; do python bench.py --forward-variant $v --no-cpu-baseline --no-lbs --lasr-iters 0 > $O/r02b_bench_v$v.json 2> $O/r02b_bench_v$v.err; done
cat $O/r02b_pytest.txt; for v in 0 1; do python -c "import json;d=json.load(open('$O/r02b_bench_v$v.json'));print($v, d['value'], d['roofline']['all_kernels_avg_ms'], d.get('relaxed_forward_math'))"; done
