"""The host-side mirror exposes the reference's operator API: same public names and the same call signatures (parameter
names, order, defaults).  tests/golden/api_surface.json was captured by oracle/gen_golden.py from the imported reference."""
import inspect
import json
import os

import lasr_amd.soft_renderer as sr
import lasr_amd.soft_renderer.functional as srf
from lasr_amd.nnutils import geom_utils, loss_utils

API = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'api_surface.json')))
MODS = {'soft_renderer': sr, 'soft_renderer.functional': srf, 'nnutils.geom_utils': geom_utils,
        'nnutils.loss_utils': loss_utils, 'ext_nnutils.loss_utils': loss_utils}
# deliberate differences (documented in DESIGN.md / the docstrings)
ALLOWED = {
    'soft_renderer.functional.look': {'up'},                 # the reference's default (None) is dereferenced and cannot run
    'soft_renderer.functional.load_obj': {'device'},         # extra keyword (target device)
}


def sig(obj):
    target = obj.__init__ if inspect.isclass(obj) else obj
    params = list(inspect.signature(target).parameters.values())
    if inspect.isclass(obj):
        params = params[1:]
    return [[p.name, None if p.default is inspect.Parameter.empty else repr(p.default)] for p in params]


def test_public_names_and_signatures_match_the_reference():
    problems = []
    for key, ref in API.items():
        if key.endswith('.methods'):
            continue
        mod, name = key.rsplit('.', 1)
        obj = getattr(MODS[mod], name, None)
        if obj is None:
            problems.append('%s missing' % key)
            continue
        mine = sig(obj)
        skip = ALLOWED.get(key, set())
        a = [p for p in mine if p[0] not in skip]
        b = [p for p in ref if p[0] not in skip]
        if [p[0] for p in a] != [p[0] for p in b]:
            problems.append('%s parameters %s != %s' % (key, [p[0] for p in a], [p[0] for p in b]))
            continue
        for (n, da), (_, db) in zip(a, b):
            if da != db and not (da is not None and db is not None and da.replace(' ', '') == db.replace(' ', '')):
                problems.append('%s default of %s: %s != %s' % (key, n, da, db))
    assert not problems, '\n'.join(problems)


def test_renderer_and_mesh_members():
    mine = {n for n in vars(sr.SoftRenderer) if not n.startswith('_')}
    assert set(API['soft_renderer.SoftRenderer.methods']) <= mine
    mesh = {n for n in dir(sr.Mesh) if not n.startswith('_')}
    assert set(API['soft_renderer.Mesh.methods']) - {'voxelize'} <= mesh      # voxelisation: never called by LASR
