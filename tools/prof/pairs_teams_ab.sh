# pair walk with one team / two teams of four waves per 16x16 tile, forced at every size, against the default choice: bash tools/prof/pairs_teams_ab.sh
R=$GRAFT_REPO_ROOT; cd $R
echo "== parity of the two-team kernel (every pair-walk case of the kernel-choice tests goes through it)"
LASR_SR_PAIR_TEAMS_MAX_TILES=1000000000000 python -m pytest tests/test_forward_kernel_choice_gpu.py tests/test_raster_vs_reference_gpu.py -m gpu -q 2>&1 | tail -2
for rep in 1 2; do for v in "default" "one" "two"; do
  unset LASR_SR_PAIR_MIN_TILES LASR_SR_PAIR_TEAMS_MAX_TILES
  [ $v = one ] && export LASR_SR_PAIR_MIN_TILES=0 LASR_SR_PAIR_TEAMS_MAX_TILES=0
  [ $v = two ] && export LASR_SR_PAIR_MIN_TILES=0 LASR_SR_PAIR_TEAMS_MAX_TILES=1000000000000
  for args in "--frames 1" "--frames 2" "--frames 4" "--frames 8" "--frames 16" "--frames 32" "--frames 64" "--frames 256" "--image-size 512 --frames 4" "--image-size 512 --frames 16"; do
    python bench.py $args --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0 --no-step-profile --steps 40 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());k=d['roofline']['all_kernels_avg_ms'];print('teams=$v', '$args', 'forward %.4f' % k['sr_forward_kernel'], 'step %.4f' % d['ms_per_step'])"
  done; done; done
