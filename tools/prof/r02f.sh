R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_raster_parity_gpu.py tests/test_manual_dp_gpu.py tests/test_softras_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -6
for i in 1 2; do python bench.py --no-cpu-baseline --no-lbs --lasr-iters 0 --steps 40 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']), d['ms_per_step'], d['roofline']['all_kernels_avg_ms'])"; done
