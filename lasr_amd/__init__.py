"""lasr_amd -- MI355X-native differentiable-rendering inner loop for LASR.

Package layout (only what the hot path of SURVEY.md section 8 needs):
  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/lasr_sr.h)
  _lib.py          ctypes binding of liblasr_hip.so (fails loudly when it is missing)
  soft_renderer/   host-side mirror of the reference `soft_renderer` operator API
  nnutils/         host-side mirror of the LASR geometry / loss helpers on the path
  synth.py         deterministic synthetic workloads (no datasets on the GPU box)
"""
__version__ = '0.1.0'
