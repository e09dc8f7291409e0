"""Pre-raster pipeline of the mirror package vs fixtures captured from the imported reference
(oracle/gen_golden.py): what reaches soft_rasterize after lighting + camera + gathers."""
import os

import numpy as np
import pytest
import torch

import lasr_amd.soft_renderer as sr
import lasr_amd.soft_renderer.rasterizer as sr_rast

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'softras_pre_raster.npz'))


@pytest.fixture
def capture(monkeypatch):
    got = {}

    def fake(face_vertices, textures, *a, **k):
        got['fv'], got['ft'], got['args'] = face_vertices.detach().numpy(), textures.detach().numpy(), a
        return torch.zeros(face_vertices.shape[0], 4, 4, 4)
    monkeypatch.setattr(sr_rast.srf, 'soft_rasterize', fake)
    return got


def t(name):
    return torch.from_numpy(GOLD[name])


def test_lasr_renderer_configuration(capture):
    r = sr.SoftRenderer(image_size=32, sigma_val=1e-4, gamma_val=1e-2, camera_mode='look_at', perspective=False,
                        aggr_func_rgb='softmax', light_mode='vertex', light_intensity_ambient=1.,
                        light_intensity_directionals=0.)
    np.testing.assert_array_equal(np.asarray(r.transform.transformer._eye, np.float32), GOLD['lasr_eye'])
    r.render_mesh(sr.Mesh(t('lasr_vpre').clone(), t('faces'), textures=t('vtex'), texture_type='vertex'))
    np.testing.assert_array_equal(capture['fv'], GOLD['lasr_fv'])      # incl. the (z+e)-e round trip
    np.testing.assert_array_equal(capture['ft'], GOLD['lasr_ft'])
    # argument order handed to the operator (rasterizer.py:45-50 of the reference)
    assert capture['args'][0] == 32 and capture['args'][6] == 1e-4 and capture['args'][7] == 'euclidean'


def test_default_lighting_perspective_vertex(capture):
    r = sr.SoftRenderer(image_size=32, camera_mode='look_at', perspective=True, viewing_angle=30, light_mode='vertex')
    v = t('verts') - torch.tensor([0, 0, 3.0])
    r.render_mesh(sr.Mesh(v, t('faces'), textures=t('vtex'), texture_type='vertex'))
    np.testing.assert_allclose(capture['fv'], GOLD['persp_vertex_fv'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(capture['ft'], GOLD['persp_vertex_ft'], rtol=0, atol=1e-6)


def test_surface_lighting_elevated_eye(capture):
    eye = sr.functional.get_points_from_angles(2.732, 30., 40.)
    np.testing.assert_allclose(np.asarray(eye, np.float32), GOLD['points_from_angles'], rtol=1e-7)
    r = sr.SoftRenderer(image_size=32, camera_mode='look_at', perspective=True, viewing_angle=25, light_mode='surface',
                        light_directions=[0.3, 0.8, -0.5], eye=list(eye))
    v = t('verts') - torch.tensor([0, 0, 3.0])
    r.render_mesh(sr.Mesh(v, t('faces'), textures=t('stex'), texture_type='surface'))
    np.testing.assert_allclose(capture['fv'], GOLD['look_surface_fv'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(capture['ft'], GOLD['look_surface_ft'], rtol=0, atol=1e-6)


def test_gathers_and_normals():
    m = sr.Mesh(t('verts'), t('faces'), textures=t('vtex'), texture_type='vertex')
    np.testing.assert_array_equal(m.face_vertices.numpy(), GOLD['face_vertices'])
    np.testing.assert_allclose(m.vertex_normals.numpy(), GOLD['vertex_normals'], atol=1e-6)
    np.testing.assert_allclose(m.surface_normals.numpy(), GOLD['surface_normals'], atol=1e-6)


def test_face_vertices_gradient_is_a_scatter_add():
    v = t('verts').clone().requires_grad_(True)
    fv = sr.functional.face_vertices(v, t('faces'))
    g = torch.randn_like(fv)
    fv.backward(g)
    ref = torch.zeros_like(v)
    for b in range(v.shape[0]):
        ref[b].index_add_(0, t('faces')[b].reshape(-1), g[b].reshape(-1, 3))
    np.testing.assert_allclose(v.grad.numpy(), ref.numpy(), atol=1e-5)


def test_api_errors_match_the_reference():
    with pytest.raises(ValueError):
        sr.SoftRasterizer(dist_func='manhattan')
    with pytest.raises(ValueError):
        sr.Lighting(light_mode='pixel')
    with pytest.raises(ValueError):
        sr.Transform(camera_mode='fisheye')
    with pytest.raises(ValueError):
        sr.Transform(camera_mode='projection')          # needs P
    r = sr.SoftRenderer(camera_mode='look_at')
    r.set_sigma(3e-5); r.set_gamma(2e-3)
    assert r.rasterizer.sigma_val == 3e-5 and r.rasterizer.gamma_val == 2e-3
    r.transform.set_eyes([0, 0, -3.0])
    assert r.transform.eyes == [0, 0, -3.0]


def test_obj_round_trip(tmp_path):
    m = sr.Mesh(t('verts')[:1], t('faces')[:1], textures=t('vtex')[:1], texture_type='vertex')
    p = str(tmp_path / 'm.obj')
    m.save_obj(p, save_texture=True)
    v, f, c = sr.functional.load_obj(p, load_texture=True, texture_type='vertex', device='cpu')
    np.testing.assert_allclose(v.numpy(), GOLD['verts'][0], atol=1e-6)
    np.testing.assert_array_equal(f.numpy(), GOLD['faces'][0])
    np.testing.assert_allclose(c.numpy(), GOLD['vtex'][0], atol=1e-6)
