"""The composed pre-raster pipeline (lighting + camera + gathers of render_mesh) on CUDA tensors -- the branch that runs
the HIP face gather (soft_renderer/functional/geometry.py) -- against the same fixtures captured from the imported reference
that tests/test_softras_pipeline.py checks on CPU tensors (oracle/gen_golden.py -> tests/golden/softras_pre_raster.npz)."""
import os

import numpy as np
import pytest
import torch

import lasr_amd.soft_renderer as sr
import lasr_amd.soft_renderer.rasterizer as sr_rast

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'softras_pre_raster.npz'))


@pytest.fixture
def capture(monkeypatch):
    got = {}

    def fake(face_vertices, textures, *a, **k):
        assert face_vertices.is_cuda and textures.is_cuda
        got['fv'], got['ft'], got['args'] = face_vertices.detach().cpu().numpy(), textures.detach().cpu().numpy(), a
        return torch.zeros(face_vertices.shape[0], 4, 4, 4, device=face_vertices.device)
    monkeypatch.setattr(sr_rast.srf, 'soft_rasterize', fake)
    return got


def t(name, dev):
    return torch.from_numpy(GOLD[name]).to(dev)


def test_lasr_renderer_configuration_cuda(capture, cuda):
    r = sr.SoftRenderer(image_size=32, sigma_val=1e-4, gamma_val=1e-2, camera_mode='look_at', perspective=False,
                        aggr_func_rgb='softmax', light_mode='vertex', light_intensity_ambient=1.,
                        light_intensity_directionals=0.).to(cuda)
    r.render_mesh(sr.Mesh(t('lasr_vpre', cuda).clone(), t('faces', cuda), textures=t('vtex', cuda), texture_type='vertex'))
    np.testing.assert_array_equal(capture['fv'], GOLD['lasr_fv'])      # bit-exact, incl. the (z+e)-e round trip
    np.testing.assert_array_equal(capture['ft'], GOLD['lasr_ft'])


def test_default_lighting_perspective_vertex_cuda(capture, cuda):
    r = sr.SoftRenderer(image_size=32, camera_mode='look_at', perspective=True, viewing_angle=30, light_mode='vertex').to(cuda)
    v = t('verts', cuda) - torch.tensor([0, 0, 3.0], device=cuda)
    r.render_mesh(sr.Mesh(v, t('faces', cuda), textures=t('vtex', cuda), texture_type='vertex'))
    np.testing.assert_allclose(capture['fv'], GOLD['persp_vertex_fv'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(capture['ft'], GOLD['persp_vertex_ft'], rtol=0, atol=1e-6)


def test_surface_lighting_elevated_eye_cuda(capture, cuda):
    eye = sr.functional.get_points_from_angles(2.732, 30., 40.)
    r = sr.SoftRenderer(image_size=32, camera_mode='look_at', perspective=True, viewing_angle=25, light_mode='surface',
                        light_directions=[0.3, 0.8, -0.5], eye=list(eye)).to(cuda)
    v = t('verts', cuda) - torch.tensor([0, 0, 3.0], device=cuda)
    r.render_mesh(sr.Mesh(v, t('faces', cuda), textures=t('stex', cuda), texture_type='surface'))
    np.testing.assert_allclose(capture['fv'], GOLD['look_surface_fv'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(capture['ft'], GOLD['look_surface_ft'], rtol=0, atol=1e-6)


def test_gathers_normals_and_gather_gradient_cuda(cuda):
    m = sr.Mesh(t('verts', cuda), t('faces', cuda), textures=t('vtex', cuda), texture_type='vertex')
    np.testing.assert_array_equal(m.face_vertices.cpu().numpy(), GOLD['face_vertices'])
    np.testing.assert_allclose(m.vertex_normals.cpu().numpy(), GOLD['vertex_normals'], atol=1e-6)
    np.testing.assert_allclose(m.surface_normals.cpu().numpy(), GOLD['surface_normals'], atol=1e-6)
    v = t('verts', cuda).clone().requires_grad_(True)
    fv = sr.functional.face_vertices(v, t('faces', cuda))
    g = torch.randn(fv.shape, generator=torch.Generator().manual_seed(0)).to(cuda)
    fv.backward(g)
    ref = torch.zeros_like(v)
    for b in range(v.shape[0]):
        ref[b].index_add_(0, t('faces', cuda)[b].reshape(-1), g[b].reshape(-1, 3))
    np.testing.assert_allclose(v.grad.cpu().numpy(), ref.cpu().numpy(), atol=1e-5)
