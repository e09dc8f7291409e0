"""Functional API, same names as the reference's soft_renderer.functional
(/root/reference/third_party/softras/soft_renderer/functional/__init__.py)."""
from .cameras import const_tensor, get_points_from_angles, look, look_at, perspective, orthogonal, projection
from .shading import ambient_lighting, directional_lighting
from .geometry import face_vertices, vertex_normals, surface_normals
from .obj_io import load_obj, save_obj
from .soft_rasterize import (soft_rasterize, soft_rasterize_raw, SoftRasterizeFunction, set_forward_flags, forward_flags,
                             set_launch_thresholds, invalidate_records)
from .load_textures import load_textures
