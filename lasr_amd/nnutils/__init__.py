"""Host-side mirror of the LASR helpers on the hot path (reference: /root/reference/nnutils/ and
third_party/ext_nnutils/loss_utils.py), backed by the HIP kernels of include/lasr_ops.h."""
