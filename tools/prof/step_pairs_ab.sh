# the optimisation steps' renders with the default kernel choice and with the pair-walk kernel forced: bash tools/prof/step_pairs_ab.sh
R=$GRAFT_REPO_ROOT; cd $R
for v in default 0 default 0; do
  if [ $v = default ]; then unset LASR_SR_PAIR_MIN_TILES; else export LASR_SR_PAIR_MIN_TILES=$v; fi
  python bench.py --frames 16 --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 2 --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for c in ('spot3_s0','camel_s4'):
    x=d['in_scope_step'][c]; r=x['raster']
    print('pair_min_tiles=$v', c, ' '.join('%s %.2f' % (k[3:-7], v['us']) for k, v in r.items()), 'raster %.1f' % x['raster_us'], 'other %.1f/%d' % (x['other_in_scope_us'], x['other_in_scope_launches']), 'wall %.0f' % x['wall_us'])"
done
