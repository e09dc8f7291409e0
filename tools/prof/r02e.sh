R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for rb in 1 0 1 0; do python bench.py --rebuild-records $rb --no-cpu-baseline --no-lbs --lasr-iters 0 --steps 40 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('rebuild=$rb', round(d['value']), d['ms_per_step'], d['roofline']['all_kernels_avg_ms'])"; done
python bench.py --no-lbs --lasr-iters 0 --steps 5 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['cpu_baseline'])"
timeout 1200 python -m pytest tests/test_softras_pipeline_gpu.py tests/test_manual_dp_gpu.py tests/test_raster_parity_gpu.py tests/test_lasr_forward_gpu.py tests/test_bench_launch_gpu.py -m gpu -x -q 2>&1 | tail -12
