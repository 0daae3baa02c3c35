"""TEST INFRASTRUCTURE ONLY - never imported by the product (softmac_amd/).

CPU restatement (numpy) of the reference's mesh -> SDF voxeliser, /root/reference/softmac/engine/primitive/mesh.py:178-241
(`trimesh2sdf`).  The reference delegates the geometry to trimesh (third-party, not vendored, absent from this image;
requirements.txt pins trimesh==3.21.5).  Its published algorithm, restated from memory of that release and PINNED by the two
cached tables the reference tree ships (below):

  on_surface (:209-210) = trimesh.proximity.closest_point: closest point of every candidate triangle (Ericson, Real-Time
      Collision Detection 5.1.5); of the two best candidates, when their squared distances agree to tol.merge = 1e-8 and the
      sample is not on the surface, the triangle whose normal makes the most positive angle with the vector surface point ->
      sample.  Here: the same choice among ALL triangles tied at the minimum (identical when two tie; with three or more
      tied trimesh's pick depends on its r-tree's candidate order, which nothing in the reference pins).
  signed_distance (:206-207, negated by the reference: negative inside): where the sample's projection on the chosen
      triangle's plane lies inside the triangle (barycentric coordinates in [-1e-12, 1 + 1e-12]) the sign is the side of that
      plane; elsewhere ray containment (`contains_points`: a ray in the fixed direction (0.4395064455, 0.617598629942,
      0.652231566745) and one in the opposite direction, hit counts mod 2, the answer where both agree; a sample one of whose
      rays hits nothing is outside; remaining disagreements are re-cast in a random direction there - here: a third fixed
      skew direction).
  normal (:213-218) = unit normal of the chosen triangle / (1 + 1e-8).

PARITY PINNED by the reference's own cached tables: tests/golden/palm_sdf.npz and tests/golden/door_sdf.npz are the
`sdf` / `normal` arrays of the two caches shipped in the reference tree (assets/gripper/6895...c4d5,
assets/door/e7ab...561a), extracted by tools/make_fixtures.py; tests/test_voxelize.py checks this file against both - since
round 4 with every sign equal (the plane-side rule is what labels 206 samples inside the door's handle legs "outside":
two coincident faces of touching boxes, the one that faces the sample wins)."""
import numpy as np


def _closest_points(p, a, b, c):
    """p (P,1,3); a,b,c (1,T,3) -> closest points (P,T,3).  Region cascade of RTCD 5.1.5."""
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        denom = 1.0 / (va + vb + vc)
        out = a + ab * (vb * denom)[..., None] + ac * (vc * denom)[..., None]                       # face interior
        m = (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0)
        w = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        out = np.where(m[..., None], b + (c - b) * w[..., None], out)                                # edge bc
        m = (vb <= 0) & (d2 >= 0) & (d6 <= 0)
        out = np.where(m[..., None], a + ac * (d2 / (d2 - d6))[..., None], out)                      # edge ac
        m = (d6 >= 0) & (d5 <= d6)
        out = np.where(m[..., None], c + 0 * out, out)                                               # vertex c
        m = (vc <= 0) & (d1 >= 0) & (d3 <= 0)
        out = np.where(m[..., None], a + ab * (d1 / (d1 - d3))[..., None], out)                      # edge ab
        m = (d3 >= 0) & (d4 <= d3)
        out = np.where(m[..., None], b + 0 * out, out)                                               # vertex b
        m = (d1 <= 0) & (d2 <= 0)
        out = np.where(m[..., None], a + 0 * out, out)                                               # vertex a
    return out


def _inside_by_ray(p, a, b, c, direction):
    """Moeller-Trumbore crossings of the ray p + t d, t > 0, with every triangle; odd count = inside."""
    e1, e2 = b - a, c - a
    h = np.cross(direction, e2)
    det = (e1 * h).sum(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / det
        s = p - a
        u = (s * h).sum(-1) * inv
        q = np.cross(s, e1)
        v = (q * direction).sum(-1) * inv
        t = (q * e2).sum(-1) * inv
    hit = (np.abs(det) > 1e-14) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0)
    return (hit.sum(-1) % 2) == 1


TOL_MERGE, TOL_ZERO = 1e-8, 1e-12
RAY_DIRECTION = np.array([0.4395064455, 0.617598629942, 0.652231566745])


def _ray_hits(p, a, b, c, direction):
    """number of triangles the ray p + t d, t > -1e-6, crosses (plane intersection inside the triangle to tol.zero)"""
    e1, e2 = b - a, c - a
    h = np.cross(direction, e2)
    det = (e1 * h).sum(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / det
        s = p - a
        u = (s * h).sum(-1) * inv
        q = np.cross(s, e1)
        v = (q * direction).sum(-1) * inv
        t = (q * e2).sum(-1) * inv
    hit = (np.abs(det) > 1e-14) & (u >= -TOL_ZERO) & (v >= -TOL_ZERO) & (u + v <= 1 + TOL_ZERO) & (t > -1e-6)
    return hit.sum(-1)


def _contains(p, a, b, c):
    """trimesh.ray.ray_util.contains_points on samples p (P,1,3)"""
    fwd, bwd = _ray_hits(p, a, b, c, RAY_DIRECTION), _ray_hits(p, a, b, c, -RAY_DIRECTION)
    cf, cb = fwd % 2 == 1, bwd % 2 == 1
    agree = cf == cb
    inside = np.where(agree, cf, False)
    free = (fwd == 0) | (bwd == 0)
    broken = ~agree & ~free
    if broken.any():
        inside[broken] = _inside_by_ray(p[broken], a, b, c, np.array([0.8017837257372732, 0.5345224838248488, 0.2672612419124244]))
    return inside


def _barycentric(ta, tb, tc, p):                             # trimesh.triangles.points_to_barycentric (Cramer)
    e0, e1, w = tb - ta, tc - ta, p - ta
    d00, d01, d11, d20, d21 = (e0 * e0).sum(-1), (e0 * e1).sum(-1), (e1 * e1).sum(-1), (w * e0).sum(-1), (w * e1).sum(-1)
    inv = 1.0 / (d00 * d11 - d01 * d01)
    v, u = (d11 * d20 - d01 * d21) * inv, (d00 * d21 - d01 * d20) * inv
    return np.stack([1 - v - u, v, u], -1)


def closest_triangle(points, vertices, faces, chunk=2048):
    """(squared distance, chosen triangle, closest point on it, number of triangles tied at the minimum) of every sample"""
    a, b, c = (vertices[faces[:, k]][None] for k in range(3))
    fn = np.cross(vertices[faces[:, 1]] - vertices[faces[:, 0]], vertices[faces[:, 2]] - vertices[faces[:, 0]])
    fn = fn / np.linalg.norm(fn, axis=1, keepdims=True)
    best = np.empty(len(points)); tid = np.empty(len(points), dtype=np.int64); ties = np.empty(len(points), dtype=np.int64)
    close = np.empty((len(points), 3))
    for s in range(0, len(points), chunk):
        p = points[s:s + chunk, None, :]
        cp = _closest_points(p, a, b, c)
        vec = p - cp
        d2 = (vec ** 2).sum(-1)
        m = d2.min(1, keepdims=True)
        cand = d2 <= m + TOL_MERGE
        with np.errstate(divide="ignore", invalid="ignore"):
            dots = ((vec / np.sqrt(d2)[..., None]) * fn[None]).sum(-1)
        dots = np.where(cand, np.where(m > TOL_MERGE, dots, 0.0), -np.inf)
        t = dots.argmax(1)                                    # (first maximum: the lowest face index among equal angles)
        r = np.arange(len(t))
        best[s:s + chunk], tid[s:s + chunk], ties[s:s + chunk], close[s:s + chunk] = m[:, 0], t, cand.sum(1), cp[r, t]
    return best, tid, close, ties


def sdf_at(pts, vertices, faces, chunk=2048):
    """signed distance (negative inside), closest-triangle normal / (1 + 1e-8) and the number of triangles tied at the minimum distance, at arbitrary points"""
    vertices = np.asarray(vertices, dtype=np.float64)
    faces = np.asarray(faces, dtype=np.int64)
    pts = np.asarray(pts, dtype=np.float64)
    best, tid, close, ties = closest_triangle(pts, vertices, faces, chunk)
    fn = np.cross(vertices[faces[:, 1]] - vertices[faces[:, 0]], vertices[faces[:, 2]] - vertices[faces[:, 0]])
    fn = fn / np.linalg.norm(fn, axis=1, keepdims=True)
    n = fn[tid]
    off = ((pts - close) * n).sum(-1)
    proj = pts - n * off[:, None]
    bc = _barycentric(vertices[faces[tid, 0]], vertices[faces[tid, 1]], vertices[faces[tid, 2]], proj)
    dist = np.sqrt(best)
    on_triangle = ~((bc < -TOL_ZERO) | (bc > 1 + TOL_ZERO)).any(1) & (dist > TOL_MERGE)
    sign = np.sign(off)                                       # +1: on the normal's side = outside (the reference's table is negative inside)
    rest = np.nonzero(~on_triangle)[0]
    a, b, c = (vertices[faces[:, k]][None] for k in range(3))
    for s in range(0, len(rest), chunk):
        ix = rest[s:s + chunk]
        sign[ix] = np.where(_contains(pts[ix, None, :], a, b, c), -1.0, 1.0)
    return sign * dist, n / (1.0 + 1e-8), ties


def mesh_to_sdf(vertices, faces, lower, res, dx, chunk=2048):
    """Tables at lower + (i,j,k) dx.  Returns sdf (res), normal (res,3) and `ties` (res): how many triangles are tied at the minimum distance - with
    more than two the triangle trimesh reports (hence the normal) is not pinned."""
    ax = [lower[d] + np.arange(res[d]) * dx for d in range(3)]
    pts = np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)
    sdf, normal, ties = sdf_at(pts, vertices, faces, chunk)
    shape = tuple(int(r) for r in res)
    return sdf.reshape(shape), normal.reshape(shape + (3,)), ties.reshape(shape)
