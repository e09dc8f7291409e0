# parity-test and bench several library builds: bash tools/prof/ab_many_test.sh lib1.so lib2.so ...
R=$GRAFT_REPO_ROOT; cd $R
for l in "$@"; do echo "== $l"; LASR_HIP_LIB=$R/$l timeout 900 python -m pytest tests/test_raster_parity_gpu.py -m gpu -q 2>&1 | grep -E "^E  .*Error|passed|failed|^FAILED" | head -6 | cut -c1-300; done
for rep in 1 2; do for l in "$@"; do LASR_HIP_LIB=$R/$l python bench.py --no-cpu-baseline --no-lbs --lasr-iters 0 --steps 40 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$l', round(d['value']), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['all_kernels_avg_ms'].items()}, round(d['relaxed_forward_math']['value']))"; done; done
