"""fused_ops.face_incidence (plain torch, runs anywhere): the CSR vertex -> corner lists the vertex-centric backward kernels walk --
corner ids 3 f + c grouped by vertex, ascending inside a vertex (the summation order of the scanning face-gather kernel)."""
import numpy as np
import torch

from lasr_amd import synth
from lasr_amd.nnutils import fused_ops


def brute(faces, V):
    ptr, lst = [0], []
    flat = faces.reshape(-1).tolist()
    for v in range(V):
        mine = [c for c, x in enumerate(flat) if x == v]
        lst += mine
        ptr.append(len(lst))
    return ptr, lst


def test_incidence_lists_of_spheres_and_of_meshes_with_unused_vertices():
    for level in (1, 2):
        v, f = synth.geodesic_sphere(2 ** level)
        faces = torch.from_numpy(np.asarray(f, np.int64))
        gen = torch.Generator().manual_seed(level)
        batch = torch.stack([faces, faces[torch.randperm(faces.shape[0], generator=gen)], faces.flip(1)])
        V = v.shape[0] + 3                                           # three trailing vertices no face uses
        ptr, lst = fused_ops.face_incidence(batch, V)
        assert ptr.dtype == torch.int32 and lst.dtype == torch.int32 and ptr.shape == (3, V + 1) and lst.shape == (3, faces.numel())
        for n in range(3):
            p, l = brute(batch[n], V)
            assert ptr[n].tolist() == p and lst[n].tolist() == l
        assert ptr[0, -4:].tolist() == [faces.numel()] * 4


def test_the_gather_backward_written_over_the_lists_is_the_scatter_add():
    v, f = synth.geodesic_sphere(2)
    faces = torch.from_numpy(np.asarray(f, np.int64))[None]
    V, F = v.shape[0], faces.shape[1]
    ptr, lst = fused_ops.face_incidence(faces, V)
    g = torch.randn(1, F, 3, 4, generator=torch.Generator().manual_seed(0))
    flat = g.reshape(3 * F, 4)
    got = torch.stack([flat[lst[0, ptr[0, k]:ptr[0, k + 1]].long()].sum(0) for k in range(V)])
    want = torch.zeros(V, 4).index_add_(0, faces.reshape(-1), flat)
    assert (got - want).abs().max() <= 1e-5
