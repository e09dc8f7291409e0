#!/usr/bin/env python3
"""Build the REFERENCE soft-rasteriser extension itself for the MI355X, outputs only into oracle/_ref/.

    python oracle/build_ref.py            # needs /root/reference (this container); hipcc cross-compiles

TEST INFRASTRUCTURE ONLY.  Nothing under lasr_amd/ may use what this builds.

What it does is what `pip install` of the reference's third_party/softras/setup.py does on a ROCm PyTorch:
`CUDAExtension` sources go through torch's own `torch.utils.hipify` (part of the installed PyTorch, the tool
every CUDA extension is built with on ROCm) and are compiled by hipcc against the installed ATen headers.
  sources (read where they lie): /root/reference/third_party/softras/soft_renderer/cuda/soft_rasterize_cuda.cpp
                                 /root/reference/third_party/softras/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu
                                 (and, for the rows f3 / f4 either side of the path, load_textures_cuda{.cpp,_kernel.cu}
                                 of the same directory and /root/reference/third_party/chamfer3D/{chamfer_cuda.cpp,chamfer3D.cu})
  hipify's output goes to a temporary directory that is removed afterwards (a translated copy of reference source
  is never kept, neither in the repository nor in oracle/_ref/): it renames the two CUDA runtime includes, the three
  `<<<>>>` launches and `cudaGetLastError`; not one line of the __global__ / __device__ code changes.
  One token-level edit is applied to the hipified host launcher, because torch 2.10 removed the overload the 2019
  source relies on: `AT_DISPATCH_FLOATING_TYPES(faces.type(), ...)` -> `faces.scalar_type()` (three call sites,
  K.cu:701,716,780; host code only; likewise `image.type()` in load_textures_cuda_kernel.cu:84).
No stand-in header, library or tool is written: every include resolves to the image's ROCm / PyTorch.

Two variants of the same sources:
  oracle/_ref/sr_ref.so          compiler defaults, as the reference's own build would have them (hipcc, like nvcc,
                                 contracts a*b+c into FMA by default)
  oracle/_ref/sr_ref_nofma.so    the same with -ffp-contract=off: the rounding sequence oracle/sr_oracle.c restates

  oracle/_ref/load_textures_ref.so, chamfer_3D_ref.so   the two side kernels (-ffp-contract=off)

They are Python extension modules exporting `forward_soft_rasterize` / `backward_soft_rasterize`
(soft_rasterize_cuda.cpp:135-138); oracle/sr_ref.py loads them.  oracle/_ref/ is git-ignored and travels to the
GPU box with the snapshot; /root/reference does not exist there and is not needed there.
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
REF_CUDA = '/root/reference/third_party/softras/soft_renderer/cuda'
REF_CHAMFER = '/root/reference/third_party/chamfer3D'
# output name -> (directory, (host .cpp, device .cu), the tensor whose deprecated `.type()` feeds AT_DISPATCH (or None),
#                 number of such call sites, extra hipcc flags)
TARGETS = {
    'sr_ref':           (REF_CUDA, ('soft_rasterize_cuda.cpp', 'soft_rasterize_cuda_kernel.cu'), 'faces', 3, []),
    'sr_ref_nofma':     (REF_CUDA, ('soft_rasterize_cuda.cpp', 'soft_rasterize_cuda_kernel.cu'), 'faces', 3,
                         ['-ffp-contract=off']),
    # rows f4 / f3 of SURVEY section 8: the texture-atlas sampler render_syn.py uses and chamfer3D's nearest-neighbour query
    'load_textures_ref': (REF_CUDA, ('load_textures_cuda.cpp', 'load_textures_cuda_kernel.cu'), 'image', 1,
                          ['-ffp-contract=off']),
    'chamfer_3D_ref':   (REF_CHAMFER, ('chamfer_cuda.cpp', 'chamfer3D.cu'), None, 0, ['-ffp-contract=off']),
}
VARIANTS = TARGETS


def available():
    return all(os.path.exists(os.path.join(d, s)) for d, srcs, _, _, _ in TARGETS.values() for s in srcs)


def built():
    return all(os.path.exists(os.path.join(OUT, n + '.so')) for n in TARGETS)


def build(force=False, verbose=False):
    """Returns the list of built .so paths ([] when /root/reference is absent and nothing was prebuilt)."""
    if not available():
        return [os.path.join(OUT, n + '.so') for n in VARIANTS if os.path.exists(os.path.join(OUT, n + '.so'))]
    srcs = sorted({os.path.join(d, s) for d, ss, _, _, _ in TARGETS.values() for s in ss})
    newest = max(os.path.getmtime(s) for s in srcs + [os.path.abspath(__file__)])
    if not force and built() and all(os.path.getmtime(os.path.join(OUT, n + '.so')) >= newest for n in TARGETS):
        return [os.path.join(OUT, n + '.so') for n in TARGETS]
    import sysconfig
    from torch.utils import cpp_extension as ce
    from torch.utils.hipify import hipify_python
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix='lasr_ref_build.')
    quiet = None if verbose else subprocess.DEVNULL
    try:
        for s in srcs:
            shutil.copy(s, tmp)
        res = hipify_python.hipify(project_directory=tmp, output_directory=tmp,
                                   extra_files=[os.path.join(tmp, os.path.basename(s)) for s in srcs],
                                   is_pytorch_extension=True, show_detailed=False, show_progress=False)
        inc = []
        for p in ce.include_paths(device_type='cuda') + [sysconfig.get_paths()['include']]:
            inc += ['-isystem', p]
        libdir = ce.library_paths(device_type='cuda')
        for _, (cpp, cu), tensor, sites, _ in TARGETS.values():
            hip = res[os.path.join(tmp, cu)].hipified_path
            text = open(hip).read()
            if tensor is not None and tensor + '.type()' in text:
                assert text.count(tensor + '.type()') == sites, 'unexpected reference source'
                open(hip, 'w').write(text.replace(tensor + '.type()', tensor + '.scalar_type()'))

        def one(item):
            name, (_, (cpp, cu), tensor, sites, extra) = item
            hip = res[os.path.join(tmp, cu)].hipified_path
            defs = ['-DTORCH_EXTENSION_NAME=' + name, '-DTORCH_API_INCLUDE_EXTENSION_H', '-D__HIP_PLATFORM_AMD__=1',
                    '-DUSE_ROCM=1', '-DHIPBLAS_V2', '-fPIC', '-std=c++17', '-w']
            o_host = os.path.join(tmp, name + '_host.o')
            o_dev = os.path.join(tmp, name + '_dev.o')
            host_src = res[os.path.join(tmp, cpp)].hipified_path or os.path.join(tmp, cpp)
            subprocess.check_call(['c++'] + defs + inc + ['-O2', '-c', host_src, '-o', o_host], stdout=quiet, stderr=quiet)
            subprocess.check_call(['/opt/rocm/bin/hipcc', '-DWITH_HIP'] + defs + inc +
                                  ['-DCUDA_HAS_FP16=1', '-D__HIP_NO_HALF_OPERATORS__=1', '-D__HIP_NO_HALF_CONVERSIONS__=1',
                                   '--offload-arch=gfx950', '-fno-gpu-rdc'] + extra + ['-c', hip, '-o', o_dev],
                                  stdout=quiet, stderr=quiet)
            so = os.path.join(OUT, name + '.so')
            subprocess.check_call(['c++', o_host, o_dev, '-shared'] + ['-L' + d for d in libdir] +
                                  ['-lc10', '-lc10_hip', '-ltorch_cpu', '-ltorch_hip', '-ltorch', '-ltorch_python',
                                   '-L/opt/rocm/lib', '-lamdhip64', '-o', so], stdout=quiet, stderr=quiet)
            return so

        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(len(TARGETS)) as pool:                 # the four targets compile side by side
            return list(pool.map(one, TARGETS.items()))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    os.environ.setdefault('PYTORCH_ROCM_ARCH', 'gfx950')
    print('\n'.join(build(force='--force' in sys.argv, verbose='-v' in sys.argv)) or 'reference sources absent, nothing built')
