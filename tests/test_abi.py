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
    assert h.lasr_abi_version() == 1
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


def test_product_package_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'lasr_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M) or 'sr_oracle' in src:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, 'product code references the oracle: %s' % bad
