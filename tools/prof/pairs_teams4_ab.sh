# four teams / two teams / default at small launches; the optimisation steps' nine-channel renders with one / two teams: bash tools/prof/pairs_teams4_ab.sh
R=$GRAFT_REPO_ROOT; cd $R
echo "== parity of the four-team kernel"
LASR_SR_PAIR_TEAMS4_MAX_TILES=1000000000000 python -m pytest tests/test_forward_kernel_choice_gpu.py tests/test_raster_vs_reference_gpu.py -m gpu -q 2>&1 | tail -2
for rep in 1 2; do for v in "default" "two" "four"; do
  unset LASR_SR_PAIR_MIN_TILES LASR_SR_PAIR_TEAMS_MAX_TILES LASR_SR_PAIR_TEAMS4_MAX_TILES
  [ $v = two ] && export LASR_SR_PAIR_MIN_TILES=0 LASR_SR_PAIR_TEAMS_MAX_TILES=1000000000000
  [ $v = four ] && export LASR_SR_PAIR_MIN_TILES=0 LASR_SR_PAIR_TEAMS4_MAX_TILES=1000000000000
  for args in "--frames 1" "--frames 2" "--frames 3" "--frames 4" "--frames 6" "--frames 8" "--frames 12" "--image-size 512 --frames 1" "--image-size 512 --frames 2"; do
    python bench.py $args --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0 --no-step-profile --steps 60 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());k=d['roofline']['all_kernels_avg_ms'];print('teams=$v', '$args', 'forward %.4f' % k['sr_forward_kernel'], 'step %.4f' % d['ms_per_step'])"
  done; done; done
for v in one two four; do
  unset LASR_SR_PAIR_MIN_TILES LASR_SR_PAIR_TEAMS_MAX_TILES LASR_SR_PAIR_TEAMS4_MAX_TILES
  [ $v = one ] && export LASR_SR_PAIR_TEAMS_MAX_TILES=0
  [ $v = two ] && export LASR_SR_PAIR_TEAMS_MAX_TILES=1000000000000
  [ $v = four ] && export LASR_SR_PAIR_TEAMS4_MAX_TILES=1000000000000
  python bench.py --frames 16 --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 2 --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for c in ('spot3_s0','camel_s4'):
    x=d['in_scope_step'][c]; r=x['raster']
    print('teams=$v', c, ' '.join('%s %.2f' % (k[3:-7], v['us']) for k, v in r.items()), 'raster %.1f' % x['raster_us'])"
done
