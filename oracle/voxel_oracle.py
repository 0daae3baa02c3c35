"""TEST INFRASTRUCTURE ONLY - never imported by the product (softmac_amd/).

CPU restatement (numpy) of the reference's mesh -> SDF voxeliser, /root/reference/softmac/engine/primitive/mesh.py:178-241
(`trimesh2sdf`).  The reference delegates the geometry to trimesh (third-party, not vendored, absent from this image;
the reference pins no version): `ProximityQuery.on_surface` = closest point and closest triangle of every sample
(trimesh.triangles.closest_point, i.e. Ericson's Real-Time Collision Detection 5.1.5), `signed_distance` = that distance
signed by a ray-casting containment test, positive inside, which the reference negates.  Restated here as

  distance : brute-force closest point over all triangles (same region logic), first minimum wins
  sign     : parity of the crossings of one skew ray per sample (an independent method from the product kernel's
             winding number)
  normal   : unit normal of the closest triangle / (1 + 1e-8)                      (mesh.py:213-218)

PARITY PINNED by the reference's own cached tables: tests/golden/palm_sdf.npz and tests/golden/door_sdf.npz are the
`sdf` / `normal` arrays of the two caches shipped in the reference tree (assets/gripper/6895...c4d5,
assets/door/e7ab...561a), extracted by tools/make_fixtures.py; tests/test_voxelize.py checks this file against both."""
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


def closest_triangle_distances(points, vertices, faces, chunk=2048):
    """Squared distance of every sample to every... reduced: (best d2, best triangle, gap to the runner-up)."""
    a, b, c = (vertices[faces[:, k]][None] for k in range(3))
    best = np.empty(len(points)); tid = np.empty(len(points), dtype=np.int64); gap = np.empty(len(points))
    for s in range(0, len(points), chunk):
        p = points[s:s + chunk, None, :]
        d2 = ((_closest_points(p, a, b, c) - p) ** 2).sum(-1)
        t = d2.argmin(1)
        best[s:s + chunk] = d2[np.arange(len(t)), t]
        tid[s:s + chunk] = t
        if d2.shape[1] > 1:
            part = np.partition(d2, 1, axis=1)
            gap[s:s + chunk] = np.sqrt(part[:, 1]) - np.sqrt(part[:, 0])
        else:
            gap[s:s + chunk] = np.inf
    return best, tid, gap


def mesh_to_sdf(vertices, faces, lower, res, dx, chunk=2048):
    """Tables at lower + (i,j,k) dx.  Returns sdf (res), normal (res,3) and `gap` (res): distance margin between the
    closest and the second-closest triangle - where it is ~0 the closest triangle (hence the normal) is ambiguous."""
    vertices = np.asarray(vertices, dtype=np.float64)
    faces = np.asarray(faces, dtype=np.int64)
    ax = [lower[d] + np.arange(res[d]) * dx for d in range(3)]
    pts = np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)
    best, tid, gap = closest_triangle_distances(pts, vertices, faces, chunk)
    a, b, c = (vertices[faces[:, k]][None] for k in range(3))
    direction = np.array([0.8017837257372732, 0.5345224838248488, 0.2672612419124244])      # (3,2,1)/sqrt(14)
    inside = np.empty(len(pts), dtype=bool)
    for s in range(0, len(pts), chunk):
        inside[s:s + chunk] = _inside_by_ray(pts[s:s + chunk, None, :], a, b, c, direction)
    sdf = np.where(inside, -1.0, 1.0) * np.sqrt(best)
    fn = np.cross(vertices[faces[:, 1]] - vertices[faces[:, 0]], vertices[faces[:, 2]] - vertices[faces[:, 0]])
    fn = fn / np.linalg.norm(fn, axis=1, keepdims=True)
    normal = fn[tid] / (1.0 + 1e-8)
    shape = tuple(int(r) for r in res)
    return sdf.reshape(shape), normal.reshape(shape + (3,)), gap.reshape(shape)
