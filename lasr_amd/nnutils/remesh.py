"""Re-meshing between optimisation stages without the external Manifold binaries.

The reference (/root/reference/nnutils/train_utils.py:419-428) exports the best hypothesis, makes it watertight with
`Manifold/build/manifold` and decimates it to `--n_faces` triangles with `Manifold/build/simplify`.  Those programs are
not part of the repository and cannot be fetched here.  LASR's meshes are deformed spheres (the template is an
icosphere and the regularisers keep it genus 0), so the same job -- a clean, evenly tessellated mesh of about n_faces
triangles on the learned surface -- is done by casting the vertex directions of a geodesic sphere of the matching
frequency from the centroid and taking the outermost intersection with the old surface."""
import math
import warnings

import numpy as np

from .. import synth


def ray_mesh_outermost(origin, dirs, verts, faces, return_first=False):
    """For each unit direction the largest t > 0 with origin + t * dir on the mesh (Moeller-Trumbore against every
    triangle); NaN where the ray misses.  return_first: also the smallest such t (first crossing)."""
    v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    e1, e2 = v1 - v0, v2 - v0                                   # [F,3]
    p = np.cross(dirs[:, None, :], e2[None])                     # [D,F,3]
    det = (p * e1[None]).sum(-1)
    ok = np.abs(det) > 1e-12
    inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
    tvec = (origin - v0)[None]                                   # [1,F,3]
    u = (tvec * p).sum(-1) * inv
    q = np.cross(np.broadcast_to(tvec, p.shape), e1[None])
    v = (dirs[:, None, :] * q).sum(-1) * inv
    t = (e2[None] * q).sum(-1) * inv
    eps = 1e-9
    hit = ok & (u >= -eps) & (v >= -eps) & (u + v <= 1 + eps) & (t > 1e-9)
    tmax = np.where(hit, t, -np.inf).max(1)
    if return_first:
        tmin = np.where(hit, t, np.inf).min(1)
        return np.where(np.isfinite(tmax), tmax, np.nan), np.where(np.isfinite(tmin), tmin, np.nan)
    return np.where(np.isfinite(tmax), tmax, np.nan)


def remesh_star(verts, faces, n_faces):
    """verts [V,3], faces [F,3] (numpy) -> (new_verts [V',3] float32, new_faces [F',3] int64) with F' = 20 nu^2 closest
    to n_faces.

    LIMITATION (warned about at run time): the result is the radial function of the surface seen from its centroid, outermost
    crossing.  Where the learned surface is not star-shaped from there -- a ray crosses it more than once: limbs, a tail, the
    gap between legs -- everything inside the outermost crossing is lost (webbing / collapsed concavities), and the loss
    compounds over the five hand-offs of scripts/template.sh.  The reference's Manifold + simplify pipeline preserves such
    geometry; a run that needs it should re-mesh externally and pass the result as the next stage's --model_path mesh.
    Directions that miss the surface altogether take the mean radius."""
    verts = np.asarray(verts, np.float64)
    faces = np.asarray(faces, np.int64)
    nu = max(1, int(round(math.sqrt(int(n_faces) / 20.0))))
    dirs, new_faces = synth.geodesic_sphere(nu)
    dirs = dirs.astype(np.float64)
    tri = verts[faces]
    area = np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    centroid = (tri.mean(1) * area[:, None]).sum(0) / max(area.sum(), 1e-30)
    t, t_first = ray_mesh_outermost(centroid, dirs, verts, faces, return_first=True)
    multi = np.isfinite(t) & (t - t_first > 1e-3 * np.nanmax(t))
    if multi.any():
        warnings.warn('remesh_star: the surface is not star-shaped from its centroid along %d of %d directions (up to %.0f %% '
                      'of the radius between first and last crossing); geometry inside the outermost crossing is dropped'
                      % (int(multi.sum()), len(t), 100 * float(np.nanmax((t - t_first)[multi] / t[multi]))), RuntimeWarning)
    if int(n_faces) != new_faces.shape[0]:
        warnings.warn('remesh_star: --n_faces %s snapped to %d (20 nu^2 faces of a geodesic sphere)'
                      % (n_faces, new_faces.shape[0]), RuntimeWarning)
    if np.isnan(t).any():
        warnings.warn('remesh_star: %d of %d directions miss the surface (open or self-intersecting mesh?); they take the '
                      'mean radius' % (int(np.isnan(t).sum()), len(t)), RuntimeWarning)
        fallback = np.nanmean(t) if np.isfinite(np.nanmean(t)) else np.linalg.norm(verts - centroid, axis=1).mean()
        t = np.where(np.isnan(t), fallback, t)
    return (centroid + dirs * t[:, None]).astype(np.float32), new_faces
