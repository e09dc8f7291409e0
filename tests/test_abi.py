"""The C-ABI library loads on a box without a GPU and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        src = open(h).read()
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        names |= set(re.findall(r'\b(lasr_[a-z0-9_]+)\s*\(', src))
    return names


def test_library_is_built_and_exports_every_declared_symbol():
    from lasr_amd import build, _lib
    so = build.build_hip()
    assert os.path.exists(so)
    h = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 8
    for n in sorted(names):
        assert hasattr(h, n), 'include/*.h declares %s but liblasr_hip.so does not export it' % n
    # the python binding knows every symbol too (no silent drift between header and ctypes signatures)
    assert names == set(_lib.SIGNATURES), names ^ set(_lib.SIGNATURES)


def test_no_compute_entry_points_without_gpu_but_metadata_calls_work():
    from lasr_amd import _lib
    h = _lib.lib()
    # the header's number, the library's and the binding's agree (signatures changed under a constant version 1)
    import re
    hdr = open(os.path.join(ROOT, 'include', 'lasr_sr.h')).read()
    assert h.lasr_abi_version() == _lib.ABI_VERSION == int(re.search(r'#define\s+LASR_ABI_VERSION\s+(\d+)', hdr).group(1))
    assert h.lasr_strerror(0) == b'ok'
    assert b'workspace' in h.lasr_strerror(-3)
    n = h.lasr_sr_workspace_bytes(2, 100, 3, 64)
    assert n >= 2 * 100 * (36 + 4) * 4
    assert h.lasr_sr_workspace_bytes(-1, 5, 3, 8) == 0
    assert h.lasr_prof_kernel_count() >= 3
    assert h.lasr_prof_kernel_name(1).startswith(b'sr_forward')


def test_argument_validation_happens_before_any_launch():
    # bad modes / sizes are rejected on the host side: safe to call without a device
    from lasr_amd import _lib
    h = _lib.lib()
    args = (None, None, None, None, None, None, 0)
    assert h.lasr_sr_forward(*args, 1, 1, 3, 8, 1., 2., 1e-3, 1e-4, 7, 9.2, 1e-2, 1, 2, 1, 1, None) == -2
    assert h.lasr_sr_forward(*args, -1, 1, 3, 8, 1., 2., 1e-3, 1e-4, 2, 9.2, 1e-2, 1, 2, 1, 1, None) == -1
    assert h.lasr_sr_forward(*args, 1, 1, 3, 8, 1., 2., 1e-3, 1e-4, 2, 9.2, 1e-2, 1, 2, 1, 1, None) == -1   # null buffers
    assert h.lasr_sr_forward(*args, 0, 1, 3, 8, 1., 2., 1e-3, 1e-4, 2, 9.2, 1e-2, 1, 2, 1, 1, None) == 0    # empty batch


def test_ex_entry_points_validate_flags_and_channels_on_the_host():
    from lasr_amd import _lib
    h = _lib.lib()
    tail = (1e-3, 1e-4, 2, 9.2, 1e-2, 1, 2, 1, 1)
    fwd = lambda channels, flags, dist=2: h.lasr_sr_forward_ex(None, None, None, None, None, None, 0, 0, 1, 3, channels, 8, 1., 2., None,
                                                               tail[0], tail[1], dist, *tail[3:], flags, None)
    assert fwd(3, 0) == 0 and fwd(3, _lib.SR_RELAXED_MATH) == 0 and fwd(3, _lib.SR_DEFAULT_FLAGS) == 0        # empty batch: ok
    assert fwd(3, _lib.SR_RECORDS_VALID) == -1 and fwd(3, 64) == -1                                            # not forward flags
    assert fwd(6, 0) == 0 and fwd(5, 0) == -2 and fwd(6, 0, dist=1) == -2                                      # 6 channels: LASR modes only
    bwd = lambda flags: h.lasr_sr_backward_ex(None, None, None, None, None, None, None, None, 0, 0, 1, 3, 3, 8, 1., 2., None,
                                              *tail, flags, None)
    assert bwd(0) == 0 and bwd(_lib.SR_RECORDS_VALID) == 0 and bwd(_lib.SR_RELAXED_MATH) == -1
    assert bwd(_lib.SR_GRADS_OVERWRITE) == 0 and bwd(_lib.SR_GRADS_OVERWRITE | _lib.SR_RECORDS_VALID) == 0 and bwd(16) == -1
    assert h.lasr_load_textures(None, None, None, None, 0, 5, 4, 4, None) == 0                                 # no faces
    assert h.lasr_load_textures(None, None, None, None, 3, 5, 4, 4, None) == -1 and h.lasr_load_textures(None, None, None, None, 3, 0, 4, 4, None) == -1


def test_product_package_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'lasr_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M) or 'sr_oracle' in src:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, 'product code references the oracle: %s' % bad


def test_fused_operator_argument_validation():
    # include/lasr_ops.h entry points of fused.hip reject bad sizes / null buffers on the host, before any launch
    from lasr_amd import _lib
    h = _lib.lib()
    assert h.lasr_flow_reproject_forward(None, None, None, None, None, None, None, -1, 4, None) == -1
    assert h.lasr_flow_reproject_forward(None, None, None, None, None, None, None, 2, 4, None) == -1      # null buffers
    assert h.lasr_flow_reproject_forward(None, None, None, None, None, None, None, 0, 4, None) == 0       # empty batch
    assert h.lasr_flow_reproject_scratch_floats(3, 65536) >= 3 * 32 * 4
    assert h.lasr_skin_weights_forward(None, None, None, None, None, 1, 65, 10, None) == -1               # more than 64 bones
    assert h.lasr_skin_weights_forward(None, None, None, None, None, 0, 5, 10, None) == 0
    assert h.lasr_quat_to_rotmat_forward(None, None, 0, None) == 0 and h.lasr_quat_to_rotmat_forward(None, None, 3, None) == -1
    assert h.lasr_flatten_forward(None, None, None, -2, 5, 5, None) == -1
    assert h.lasr_face_gather_forward(None, None, None, 1, 5, 5, 0, None) == -1                           # zero channels
    assert h.lasr_nearest_point(None, None, None, None, 1, 5, 0, None) == -1                              # empty target set
    assert h.lasr_point_mesh_forward(None, None, None, None, None, None, None, None, 1, 5, 0, 5, None) == -1
    assert h.lasr_point_mesh_scratch_floats(1, 2560, 1282) >= 2 * (40 * 1282 + 6 * 2560)        # 40 face chunks, 6 blocks of 256 points
    assert h.lasr_cosdist_forward(None, None, None, None, 2, 8, 16, 0, None) == -1                        # rep must be >= 1
    assert h.lasr_cosdist_scratch_floats(4, 1000) >= 16
    # unknown forward flag bits are refused before anything is launched
    assert h.lasr_sr_forward_opt(None, None, None, None, None, None, 0, 1, 1, 3, 3, 8, 1., 2., None, 1e-3, 1e-4, 2, 9.21, 1e-2,
                                 1, 2, 1, 1, None, 64, None, None) == -1


def test_header_constants_match_the_python_mirror():
    import re
    from lasr_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ops = open(os.path.join(root, 'include', 'lasr_ops.h')).read()
    sr = open(os.path.join(root, 'include', 'lasr_sr.h')).read()

    def define(text, name):
        m = re.search(r'#define\s+%s\s+\(?(-?\d+)\)?' % name, text)
        assert m, name
        return int(m.group(1))
    assert define(ops, 'LASR_MEANS_MAX_TERMS') == _lib.MEANS_MAX_TERMS
    assert define(ops, 'LASR_TAIL_MAX_GROUPS') == _lib.TAIL_MAX_GROUPS
    assert define(sr, 'LASR_SR_RELAXED_MATH') == _lib.SR_RELAXED_MATH
    assert define(sr, 'LASR_SR_RECORDS_VALID') == _lib.SR_RECORDS_VALID
    assert define(sr, 'LASR_SR_GRADS_OVERWRITE') == _lib.SR_GRADS_OVERWRITE


def test_options_struct_of_the_python_mirror_is_the_header_s():
    # lasr_sr_options (include/lasr_sr.h) <-> _lib.SrOptions: same fields, same order, all `long long`; the workspace size the
    # library asks for depends on the image size (the forward's tile-order table lives there)
    import ctypes
    import re
    from lasr_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sr = open(os.path.join(root, 'include', 'lasr_sr.h')).read()
    body = re.search(r'typedef struct lasr_sr_options \{(.*?)\} lasr_sr_options;', sr, re.S).group(1)
    fields = re.findall(r'long long\s+(\w+);', body)
    assert fields == [n for n, _ in _lib.SrOptions._fields_] == ['coop8_max_tiles', 'coop_max_tiles', 'choose_max_tiles', 'order_max_tiles',
                                                                 'pair_min_tiles']
    assert all(t is ctypes.c_longlong for _, t in _lib.SrOptions._fields_) and ctypes.sizeof(_lib.SrOptions) == 8 * len(fields)
    h = _lib.lib()
    small, large = h.lasr_sr_workspace_bytes(8, 100, 3, 64), h.lasr_sr_workspace_bytes(8, 100, 3, 256)
    assert large - small >= 8 * (32 * 32 - 8 * 8) * 4                      # 4 bytes per (image, 8x8 tile) at least
    assert h.lasr_sr_workspace_bytes(8, 100, 3, 0) <= small


def test_operator_launch_options_are_per_call_arguments_not_library_state():
    # the operator keeps its lasr_sr_options struct on the Python side and passes it with every forward call (the library has
    # no setters since round 4); no arguments = NULL = every default
    import importlib
    sr_mod = importlib.import_module('lasr_amd.soft_renderer.functional.soft_rasterize')
    from lasr_amd.soft_renderer import functional as srf
    h = __import__('lasr_amd._lib', fromlist=['lib']).lib()
    assert not hasattr(h, 'lasr_sr_set_launch_thresholds') and not hasattr(h, 'lasr_sr_set_forward_math')
    try:
        assert sr_mod._options_ref() is None
        srf.set_launch_thresholds(0, 10, 20, 0)
        o = sr_mod._launch_options
        assert (o.coop8_max_tiles, o.coop_max_tiles, o.choose_max_tiles, o.order_max_tiles) == (0, 10, 20, 0)
        assert sr_mod._options_ref() is not None
        srf.set_launch_thresholds(order_max_tiles=0)                # only the tile order switched off: the rest stay defaults
        o = sr_mod._launch_options
        assert (o.coop8_max_tiles, o.coop_max_tiles, o.choose_max_tiles, o.order_max_tiles) == (-1, -1, -1, 0)
    finally:
        srf.set_launch_thresholds()
    assert sr_mod._options_ref() is None
    old = srf.set_forward_flags(srf.SR_SEGMENTED) if hasattr(srf, 'SR_SEGMENTED') else None
    if old is not None:
        srf.set_forward_flags(old)


def test_round5_entry_points_validate_on_the_host():
    # argument checks of the entry points added in round 5 run before any launch: callable without a device
    from lasr_amd import _lib
    h = _lib.lib()
    n = None
    assert h.lasr_face_gather_backward_csr(n, n, n, 1, n, 1, 5, 4, 0, n) == -1                   # zero channels
    assert h.lasr_face_gather_backward_csr(n, n, n, 1, n, 0, 5, 4, 3, n) == 0                    # empty batch
    assert h.lasr_face_gather_backward_csr(n, n, n, 1, n, 2, 5, 4, 3, n) == -1                   # null buffers
    assert h.lasr_bone_fixup_pair_forward(n, n, n, n, n, n, n, 3, 1, 2, n) == -1                 # odd number of meshes: no frame pairs
    assert h.lasr_bone_fixup_pair_forward(n, n, n, n, n, n, n, 0, 1, 2, n) == 0
    assert h.lasr_bone_fixup_pair_backward(n, n, n, n, n, n, n, n, n, 4, 3, 2, n) == -1          # M not a multiple of H
    reg_f = lambda N, NA, V, NC, P, Q: h.lasr_step_regularisers_forward(n, n, n, n, n, n, n, n, n, n, n, n, N, NA, V, 0,      # noqa: E731
                                                                        n, n, n, n, n, NC, P, Q, n)
    assert reg_f(0, 0, 5, 0, 0, 0) == 0 and reg_f(0, 0, 5, 2, 0, 3) == -1 and reg_f(0, 0, 5, 2, 3, 3) == -1 and reg_f(-1, 0, 5, 0, 0, 0) == -1
    reg_b = lambda N, NA, V, NC, P, Q: h.lasr_step_regularisers_backward(n, n, n, n, n, n, n, n, n, n, n, n, n, n, n, n, n, N, NA, V, 0,   # noqa: E731
                                                                         n, n, n, n, n, n, n, NC, P, Q, n)
    assert reg_b(0, 0, 5, 0, 0, 0) == 0 and reg_b(0, 0, 5, 1, 2, 2) == -1 and reg_b(0, 0, 0, 0, 0, 0) == 0
    assert h.lasr_mesh_regularisers_forward(n, n, n, n, n, n, n, n, n, n, n, n, 0, 0, 5, 0, n) == 0
    assert h.lasr_cosdist_multi_forward(n, n, n, n, 0, n, n, 2, 1, n) == -1                      # no layers
    assert h.lasr_cosdist_multi_scratch_floats(n, 3, 2) == 0
    assert h.lasr_raster_faces_scratch_floats(2, 10, 16) > 0
    assert h.lasr_project_points_forward(n, n, n, n, n, n, n, 0, 1, 2, n) in (0, -1)
    assert h.lasr_render_tables_forward_imgs(n, n, n, n, 0, n, n, n, n, 1.0, n, n, n, n, n, n, n, n, n, 0, 1, 4, n) in (0, -1)
    assert h.lasr_pose_chain_forward(n, 1, *([n] * 17), 0, 1, 2, 1.0, n) == 0 and h.lasr_pose_chain_forward(n, 1, *([n] * 17), 1, 1, 2, 1.0, n) == -1
    assert h.lasr_pose_chain_backward(n, 1, *([n] * 22), 0, 1, 2, n) == 0 and h.lasr_pose_chain_backward(n, 0, *([n] * 22), 1, 1, 2, n) == -1
