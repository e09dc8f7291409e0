"""Re-meshing between optimisation stages without the external Manifold binaries.

The reference (/root/reference/nnutils/train_utils.py:419-428) exports the best hypothesis, makes it watertight with
`Manifold/build/manifold` (a dense, ~10^4-resolution re-sampling of the surface) and decimates that to EXACTLY `--n_faces`
triangles with `Manifold/build/simplify -m` (manifold-preserving quadric edge collapse).  Those programs are not part of the
repository and cannot be fetched here.  LASR's meshes are closed 2-manifolds already (the template is an icosphere and every
hand-off starts from the previous stage's mesh), so the same job is done in place:

    remesh_exact : 1 -> 4 midpoint subdivision until the mesh is at least 1.5 times as fine as requested (the dense re-sampling;
                   every new vertex lies on the learned surface), then quadric-error edge collapses -- link condition and
                   normal-flip tests keep the surface a manifold of the same genus, limbs included -- down to exactly
                   n_faces triangles.  A closed manifold has an even face count and each collapse removes two faces, so every
                   even n_faces >= 4 is reachable; scripts/template.sh's 1600 / 1920 / 2240 / 2560 / 2880 give exactly that.

`remesh_star` (rounds 1-2: radial re-sampling onto a geodesic sphere, face count snapped to 20 nu^2, star-shaped surfaces
only) is kept for callers that want a sphere-parameterised result; the trainer no longer uses it."""
import heapq
import math
import warnings

import numpy as np

from .. import synth


# --------------------------------------------------------------------------------------------------------------------
# exact-count, topology-preserving re-mesher
# --------------------------------------------------------------------------------------------------------------------
def _subdivide(verts, faces):
    """1 -> 4 midpoint subdivision; new vertices are edge midpoints (on the piecewise-linear surface)."""
    key = {}
    new = []
    nv = len(verts)

    def mid(a, b):
        k = (a, b) if a < b else (b, a)
        if k not in key:
            key[k] = nv + len(new)
            new.append(0.5 * (verts[a] + verts[b]))
        return key[k]

    out = []
    for a, b, c in faces:
        ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
        out += [(a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca)]
    return np.vstack([verts, np.asarray(new)]), out


def _is_closed_manifold(faces):
    count = {}
    for a, b, c in faces:
        for e in ((a, b), (b, c), (c, a)):
            if e in count:
                return False                      # the same directed edge twice: inconsistent orientation / non-manifold
            count[e] = 1
    return all((b, a) in count for (a, b) in count)


# The collapse order decides the result, and it hangs on comparisons of double-precision costs: the few small dense operations
# are written out in plain arithmetic so that no BLAS / LAPACK build (which pick kernels -- FMA or not, blocking -- by CPU) has
# a say, and the same input gives the same mesh on every machine.
def _dot3(a, b):
    return float(a[0]) * float(b[0]) + float(a[1]) * float(b[1]) + float(a[2]) * float(b[2])


def _norm3(a):
    return math.sqrt(_dot3(a, a))


def _det3(m):
    a, b, c, d, e, f, g, h, i = (float(x) for x in (m[0][0], m[0][1], m[0][2], m[1][0], m[1][1], m[1][2], m[2][0], m[2][1], m[2][2]))
    return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)


def _solve3(m, r, det):
    """Cramer's rule for the symmetric 3x3 system of the optimal collapse point (det = _det3(m), already known non-zero)."""
    cols = [[float(m[k][j]) for k in range(3)] for j in range(3)]
    rhs = [float(x) for x in r]
    out = []
    for j in range(3):
        mj = [[rhs[k] if jj == j else cols[jj][k] for jj in range(3)] for k in range(3)]
        out.append(_det3(mj) / det)
    return np.asarray(out)


def _quadric(q, p):
    """[p, 1] q [p, 1]^T summed in a fixed order."""
    h = (float(p[0]), float(p[1]), float(p[2]), 1.0)
    total = 0.0
    for i in range(4):
        row = 0.0
        for j in range(4):
            row += float(q[i][j]) * h[j]
        total += h[i] * row
    return total


def remesh_exact(verts, faces, n_faces, reg=1e-3, sliver=0.1):
    """verts [V,3], faces [F,3] (numpy; a closed, consistently oriented 2-manifold) -> (new_verts [V',3] float32,
    new_faces [n_faces,3] int64): the same surface (same genus, same orientation) with exactly n_faces triangles.

    Edge collapse cost = Garland-Heckbert quadric error (area-weighted face planes of the dense mesh) at the best of
    {optimal point, midpoint, end points} + reg * |edge|^2 * mean face area (keeps flat regions evenly tessellated, where the
    quadric alone is zero).  A collapse is rejected when the one-rings of its end points share anything but the two
    opposite vertices (link condition: the result would pinch), when a surviving triangle's normal would turn by more
    than ~80 degrees, or when it would create a sliver."""
    n_faces = int(n_faces)
    V = np.asarray(verts, np.float64).copy()
    F = [tuple(int(x) for x in f) for f in np.asarray(faces)]
    if n_faces < 4 or n_faces % 2:
        raise ValueError('remesh_exact: a closed triangle mesh has an even number of faces >= 4 (got --n_faces %d)' % n_faces)
    if not _is_closed_manifold(F):
        raise ValueError('remesh_exact: the input is not a closed, consistently oriented manifold')
    while 2 * len(F) < 3 * n_faces:
        V, F = _subdivide(V, F)
    if len(F) == n_faces:
        return V.astype(np.float32), np.asarray(F, np.int64)

    nv = len(V)
    faces_l = [list(f) for f in F]                       # None once removed
    vf = [set() for _ in range(nv)]                      # vertex -> incident face ids
    for i, f in enumerate(faces_l):
        for v in f:
            vf[v].add(i)
    tri = V[np.asarray(F)]
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    area2 = np.linalg.norm(nrm, axis=1)
    mean_area = 0.5 * float(area2.mean())
    unit = nrm / np.maximum(area2, 1e-30)[:, None]
    plane = np.concatenate([unit, -(unit * tri[:, 0]).sum(1, keepdims=True)], 1)            # [F,4]
    Kf = plane[:, :, None] * plane[:, None, :] * (0.5 * area2)[:, None, None]
    Q = np.zeros((nv, 4, 4))
    for i, f in enumerate(F):
        for v in f:
            Q[v] += Kf[i]
    stamp = [0] * nv
    alive = [True] * nv
    # sliver bar relative to the input's own triangles (a stretched limb is made of thin triangles to begin with)
    emax2 = np.max([((tri[:, i] - tri[:, (i + 1) % 3]) ** 2).sum(1) for i in range(3)], axis=0)
    sliver = min(sliver, 0.5 * float(np.percentile(area2 / np.maximum(emax2, 1e-300), 5)))

    def neighbours(v):
        out = set()
        for fi in vf[v]:
            out.update(faces_l[fi])
        out.discard(v)
        return out

    def cost(a, b):
        q = Q[a] + Q[b]
        cands = [0.5 * (V[a] + V[b]), V[a], V[b]]
        A3 = q[:3, :3]
        det = _det3(A3)
        if abs(det) > 1e-12 * max(np.abs(A3).max(), 1e-30) ** 3:
            p = _solve3(A3, -q[:3, 3], det)
            if _norm3(p - cands[0]) <= 2.0 * _norm3(V[a] - V[b]):                             # no wild extrapolation
                cands.insert(0, p)
        best, bp = None, None
        for p in cands:
            c = _quadric(q, p)
            if best is None or c < best:
                best, bp = c, p
        return max(best, 0.0) + reg * float(((V[a] - V[b]) ** 2).sum()) * mean_area, bp

    heap = []

    def push(a, b):
        if a > b:
            a, b = b, a
        c, p = cost(a, b)
        heapq.heappush(heap, (c, a, b, stamp[a], stamp[b], tuple(p)))

    def push_all():
        seen = set()
        for f in faces_l:
            if f is None:
                continue
            for a, b in ((f[0], f[1]), (f[1], f[2]), (f[2], f[0])):
                k = (a, b) if a < b else (b, a)
                if k not in seen:
                    seen.add(k)
                    push(*k)

    def flips(v, other, p, shared):
        for fi in vf[v]:
            if fi in shared:
                continue
            f = faces_l[fi]
            old = [V[x] for x in f]
            new = [p if (x == v or x == other) else V[x] for x in f]
            n0 = np.cross(old[1] - old[0], old[2] - old[0])
            n1 = np.cross(new[1] - new[0], new[2] - new[0])
            l0, l1 = _norm3(n0), _norm3(n1)
            if l1 <= 1e-14 * max(l0, 1e-30) or _dot3(n0, n1) < 0.17 * l0 * l1:
                return True
            # no new slivers: 2 * area / longest edge^2 (0.87 for an equilateral triangle) may not fall below `sliver`
            # unless the triangle was at least as thin before
            q1 = l1 / max(((new[1] - new[0]) ** 2).sum(), ((new[2] - new[1]) ** 2).sum(), ((new[0] - new[2]) ** 2).sum())
            if q1 < sliver:
                q0 = l0 / max(((old[1] - old[0]) ** 2).sum(), ((old[2] - old[1]) ** 2).sum(), ((old[0] - old[2]) ** 2).sum())
                if q1 < q0:
                    return True
        return False

    push_all()
    n_alive = len(F)
    progress = True
    while n_alive > n_faces:
        if not heap:
            # every queued edge was inadmissible when it came up; edges rejected earlier may be admissible now
            if not progress:
                break
            progress = False
            push_all()
            continue
        c, a, b, sa, sb, p = heapq.heappop(heap)
        if not (alive[a] and alive[b]) or sa != stamp[a] or sb != stamp[b]:
            continue
        shared = vf[a] & vf[b]
        if len(shared) != 2:
            continue
        opp = set()
        for fi in shared:
            opp.update(x for x in faces_l[fi] if x != a and x != b)
        p = np.asarray(p)
        if len(opp) != 2 or (neighbours(a) & neighbours(b)) != opp or any(len(vf[o]) <= 3 for o in opp) \
                or flips(a, b, p, shared) or flips(b, a, p, shared):
            continue
        # collapse b into a, placed at p
        for fi in shared:
            for x in faces_l[fi]:
                vf[x].discard(fi)
            faces_l[fi] = None
        for fi in list(vf[b]):
            f = faces_l[fi]
            f[f.index(b)] = a
            vf[a].add(fi)
        vf[b] = set()
        alive[b] = False
        V[a] = p
        Q[a] = Q[a] + Q[b]
        stamp[a] += 1                                    # queued entries of a's edges are stale (position and quadric moved)
        n_alive -= 2
        progress = True
        for v in neighbours(a):
            push(a, v)
    if n_alive != n_faces:
        raise RuntimeError('remesh_exact: stopped at %d faces, %d requested (no admissible collapse left)' % (n_alive, n_faces))
    keep = [f for f in faces_l if f is not None]
    used = sorted({v for f in keep for v in f})
    index = {v: i for i, v in enumerate(used)}
    out_f = np.asarray([[index[v] for v in f] for f in keep], np.int64)
    return V[used].astype(np.float32), out_f


# --------------------------------------------------------------------------------------------------------------------
# radial re-sampling (rounds 1-2); kept as a utility
# --------------------------------------------------------------------------------------------------------------------
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
    to n_faces: the radial function of the surface seen from its centroid (outermost crossing) sampled on a geodesic sphere.
    Star-shaped surfaces only -- limbs, a tail, the gap between legs are webbed over (warned about at run time); the trainer
    uses remesh_exact instead.  Directions that miss the surface altogether take the mean radius."""
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
