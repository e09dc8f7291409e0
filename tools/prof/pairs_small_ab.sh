# small launches: pair walk forced (LASR_SR_PAIR_MIN_TILES=0) vs the default choice: bash tools/prof/pairs_small_ab.sh
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for v in default 0; do
  if [ $v = default ]; then unset LASR_SR_PAIR_MIN_TILES; else export LASR_SR_PAIR_MIN_TILES=$v; fi
  for args in "--frames 1" "--frames 2" "--frames 4" "--frames 8" "--frames 12" "--image-size 512 --frames 1" "--image-size 512 --frames 2"; do
    python bench.py $args --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0 --no-step-profile --steps 100 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('pair_min_tiles=$v', '$args', round(d['value']), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['all_kernels_avg_ms'].items()})"
  done; done; done
