"""Build the native pieces in-tree (so the .so files travel with the repo snapshot).

    python -m lasr_amd.build            # HIP library for gfx950 (cross-compiles without a GPU)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'liblasr_hip.so')


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def hip_sources():
    root = os.path.dirname(HERE)
    srcs = [os.path.join(CSRC, n) for n in sorted(os.listdir(CSRC)) if n.endswith(('.hip', '.h', 'Makefile'))]
    srcs += [os.path.join(root, 'include', n) for n in sorted(os.listdir(os.path.join(root, 'include')))]
    return srcs


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> lasr_amd/csrc/liblasr_hip.so.  Returns the path."""
    if force or _stale(LIB, hip_sources()):
        cmd = ['make', '-C', CSRC] + (['-B'] if force else [])
        out = None if verbose else subprocess.DEVNULL
        subprocess.check_call(cmd, stdout=out)
    return LIB


if __name__ == '__main__':
    print(build_hip(force='--force' in sys.argv, verbose=True))
