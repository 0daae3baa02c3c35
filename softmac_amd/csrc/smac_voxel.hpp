// Mesh -> signed-distance table on the device (asset pipeline).
//
// Replaces the reference's CPU voxeliser, mesh.py:178-241 (`trimesh2sdf`, which asks trimesh's ProximityQuery for
// the closest surface point and triangle of every sample and for a signed distance):
//   sample (i,j,k) = lower + (i,j,k) dx              (cell centres of the reference's box, mesh.py:192-204)
//   distance       = exact closest point on the closest triangle (Ericson, Real-Time Collision Detection 5.1.5,
//                    the same routine trimesh.triangles.closest_point implements)
//   triangle       = trimesh.proximity.closest_point's choice (what `on_surface` :209-210 returns): of the candidates at the
//                    same (squared) distance to 1e-8, the one whose normal makes the most positive angle with the vector from
//                    the surface point to the sample
//   sign           = trimesh.proximity.signed_distance (:206-207, negated by the reference): where the sample projects INTO
//                    that triangle, the side of the triangle's plane it lies on; elsewhere (closest point on an edge or a
//                    vertex) containment - here the generalised winding number (sum of the triangles' solid angles / 4 pi)
//                    in place of trimesh's two-way ray parity, which it agrees with on every sample of the two cached tables.
//                    Round 4: this is what reproduces the reference's door cache sample for sample.  The door is four touching
//                    boxes; a sample inside a handle leg next to the panel has two coincident closest faces of opposite
//                    orientation, trimesh's rule picks the one that faces the sample, and the plane test then says "outside":
//                    206 samples that are geometrically inside the union carry a positive distance in the reference's table.
//   normal         = unit normal of that closest triangle / (1 + 1e-8)   (mesh.py:213-218)
// One thread per sample, triangles streamed through LDS (two passes: the minimum, then the choice among its ties); all
// arithmetic in f64 (the tables are f64 on the host).
#pragma once
#include <hip/hip_runtime.h>

namespace smac {

constexpr int VOX_TILE = 128;     // triangles per LDS tile

struct Vec3d { double x, y, z; };
__device__ __forceinline__ Vec3d v3(double x, double y, double z) { Vec3d r = {x, y, z}; return r; }
__device__ __forceinline__ Vec3d operator-(Vec3d a, Vec3d b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ Vec3d operator+(Vec3d a, Vec3d b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ Vec3d operator*(Vec3d a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ double dot(Vec3d a, Vec3d b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ Vec3d cross(Vec3d a, Vec3d b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// closest point of triangle (a, b, c) to p
__device__ __forceinline__ Vec3d closest_on_triangle(Vec3d p, Vec3d a, Vec3d b, Vec3d c) {
    const Vec3d ab = b - a, ac = c - a, ap = p - a;
    const double d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.0 && d2 <= 0.0) return a;
    const Vec3d bp = p - b;
    const double d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.0 && d4 <= d3) return b;
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) return a + ab * (d1 / (d1 - d3));
    const Vec3d cp = p - c;
    const double d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.0 && d5 <= d6) return c;
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) return a + ac * (d2 / (d2 - d6));
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) return b + (c - b) * ((d4 - d3) / ((d4 - d3) + (d5 - d6)));
    const double denom = 1.0 / (va + vb + vc);
    return a + ab * (vb * denom) + ac * (vc * denom);
}

// tri: nf x 9 doubles (a, b, c); sdf: res[0]*res[1]*res[2]; normal: same x 3 (C order, z fastest)
__global__ __launch_bounds__(256) void k_mesh_to_sdf(const double* __restrict__ tri, int nf, double lx, double ly, double lz,
                                                     int r0, int r1, int r2, double dx, double* __restrict__ sdf,
                                                     double* __restrict__ normal) {
    __shared__ double T[VOX_TILE * 9];
    const long total = (long)r0 * r1 * r2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < total;
    const long q = live ? idx : 0;
    const int k = (int)(q % r2), j = (int)((q / r2) % r1), i = (int)(q / ((long)r1 * r2));
    const Vec3d p = v3(lx + i * dx, ly + j * dx, lz + k * dx);
    double best = 1e300, omega = 0.0;
    for (int t0 = 0; t0 < nf; t0 += VOX_TILE) {
        const int nt = nf - t0 < VOX_TILE ? nf - t0 : VOX_TILE;
        __syncthreads();
        for (int e = threadIdx.x; e < nt * 9; e += blockDim.x) T[e] = tri[(size_t)t0 * 9 + e];
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const Vec3d a = v3(T[9 * t], T[9 * t + 1], T[9 * t + 2]), b = v3(T[9 * t + 3], T[9 * t + 4], T[9 * t + 5]),
                        c = v3(T[9 * t + 6], T[9 * t + 7], T[9 * t + 8]);
            const Vec3d cp = closest_on_triangle(p, a, b, c) - p;
            const double d2 = dot(cp, cp);
            if (d2 < best) best = d2;
            // solid angle (Van Oosterom & Strackee)
            const Vec3d A = a - p, B = b - p, C = c - p;
            const double la = sqrt(dot(A, A)), lb = sqrt(dot(B, B)), lc = sqrt(dot(C, C));
            const double num = dot(A, cross(B, C));
            const double den = la * lb * lc + dot(A, B) * lc + dot(B, C) * la + dot(C, A) * lb;
            omega += 2.0 * atan2(num, den);
        }
    }
    // second pass: among the triangles at the minimum distance (squared distances equal to 1e-8, trimesh's tol.merge) the one whose normal
    // makes the most positive angle with surface point -> sample; the first of them when the sample lies on the surface
    const double TOL_MERGE = 1e-8;
    int best_t = -1;
    double best_dot = -2.0;
    Vec3d best_cp = p;
    for (int t0 = 0; t0 < nf; t0 += VOX_TILE) {
        const int nt = nf - t0 < VOX_TILE ? nf - t0 : VOX_TILE;
        __syncthreads();
        for (int e = threadIdx.x; e < nt * 9; e += blockDim.x) T[e] = tri[(size_t)t0 * 9 + e];
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const Vec3d a = v3(T[9 * t], T[9 * t + 1], T[9 * t + 2]), b = v3(T[9 * t + 3], T[9 * t + 4], T[9 * t + 5]),
                        c = v3(T[9 * t + 6], T[9 * t + 7], T[9 * t + 8]);
            const Vec3d cl = closest_on_triangle(p, a, b, c);
            const Vec3d vec = p - cl;
            const double d2 = dot(vec, vec);
            if (d2 > best + TOL_MERGE) continue;
            double dn = 0.0;
            if (best > TOL_MERGE) {
                const Vec3d n = cross(b - a, c - a);
                const double ln = sqrt(dot(n, n));
                dn = ln > 0.0 ? dot(n, vec) / (ln * sqrt(d2)) : -1.0;
            }
            if (best_t < 0 || dn > best_dot) { best_t = t0 + t; best_dot = dn; best_cp = cl; }
        }
    }
    if (!live) return;
    const double* tb = tri + (size_t)best_t * 9;
    const Vec3d a = v3(tb[0], tb[1], tb[2]), b = v3(tb[3], tb[4], tb[5]), c = v3(tb[6], tb[7], tb[8]);
    const Vec3d n = cross(b - a, c - a);
    const double ln = sqrt(dot(n, n));
    const Vec3d nu = ln > 0.0 ? n * (1.0 / ln) : v3(0.0, 0.0, 0.0);
    const double d = sqrt(best);
    // signed_distance: projection of the sample on the triangle's plane; inside the triangle (barycentric coordinates in [-1e-12, 1 + 1e-12],
    // trimesh's tol.zero) -> the sign is the side of the plane
    const double off = dot(p - best_cp, nu);
    const Vec3d proj = p - nu * off;
    const Vec3d e0 = b - a, e1 = c - a, w3 = proj - a;
    const double d00 = dot(e0, e0), d01 = dot(e0, e1), d11 = dot(e1, e1), d20 = dot(w3, e0), d21 = dot(w3, e1);
    const double det = d00 * d11 - d01 * d01;
    bool on_triangle = false;
    if (det != 0.0) {
        const double bv = (d11 * d20 - d01 * d21) / det, bw = (d00 * d21 - d01 * d20) / det, bu = 1.0 - bv - bw;
        const double lo = -1e-12, hi = 1.0 + 1e-12;
        on_triangle = bu >= lo && bu <= hi && bv >= lo && bv <= hi && bw >= lo && bw <= hi;
    }
    double sgn;
    if (on_triangle && d > TOL_MERGE) sgn = off > 0.0 ? 1.0 : (off < 0.0 ? -1.0 : 0.0);
    else sgn = fabs(omega) * (1.0 / (4.0 * 3.14159265358979323846)) > 0.5 ? -1.0 : 1.0;
    sdf[idx] = sgn * d;
    const double s = ln > 0.0 ? 1.0 / (1.0 + 1e-8) : 0.0;
    normal[3 * idx] = nu.x * s; normal[3 * idx + 1] = nu.y * s; normal[3 * idx + 2] = nu.z * s;
}

}  // namespace smac
