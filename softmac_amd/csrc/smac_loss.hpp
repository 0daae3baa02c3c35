// Chamfer loss on the device.
//
// Replaces the reference's O(N^2) Taichi kernels `chamfer_closest` / `compute_chamfer_loss_kernel` and the
// generated adjoint of the latter (losses/loss_pour.py:44-70, loss_grip.py:45-68):
//     L = sum_i min_j |x_i - t_j|^2  +  sum_j min_i |x_i - t_j|^2 ,    dL/dx_i = 2 (x_i - t_nn(i)) + sum_{j: nn(j) = i} 2 (x_i - t_j)
// with the reference's tie rule (first minimum in index order: strict `<` while j runs upwards).
// The nearest neighbours come from a two-level uniform grid instead of the all-pairs sweep:
//   * every point set is binned by (coarse cell of 8^3 fine cells, fine cell); one counting sort gives contiguous
//     ranges for both levels;
//   * a query walks the fine cells ring by ring (rings 0..2, stop as soon as best <= (ring * h)^2);
//   * a query that is still open (the clouds are far apart - the usual state early in an optimisation) scans the
//     coarse cells: nearest box first, then every non-empty box whose distance is below the best found.
// Exact for any configuration; 1M x 1M takes milliseconds instead of 10^12 pair tests.  Distances in f64.
#pragma once
#include <hip/hip_runtime.h>

#include "smac_math.hpp"

namespace smac {

// position rows of a frame (pos_of<R>: doubles, or 32-bit fixed point in f32 mode)
template <class R> __device__ __forceinline__ double rd_pos(const R* row, int p) {
    return pos_get(((const typename pos_of<R>::type*)row)[poff(p)]);         // `row` = frame + rowbase(c): see smac_math.hpp "Frame layout"
}

struct PointIndex {
    int n;                 // fine cells per dimension (multiple of 8)
    int npts;
    const int* cell_start; // (n/8)^3 * 512 + 1, cells ordered coarse-major
    const int* ids;        // sorted slot -> caller's index (tie rule) ...
    const int* slots;      // ... and -> storage slot (where the gradient goes)
    const double* pts;     // sorted positions, 3 per point
};

__device__ __forceinline__ int vox_key(int n, int i, int j, int k) {
    const int nc = n >> 3;
    return (((i >> 3) * nc + (j >> 3)) * nc + (k >> 3)) * 512 + (((i & 7) << 6) | ((j & 7) << 3) | (k & 7));
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ void cell_of_point(int n, const double* p, int* c) {
    for (int d = 0; d < 3; ++d) c[d] = clampi((int)floor(p[d] * n), 0, n - 1);
}

// points given as three rows (SoA frame, scalar type R) or as an (m,3) f64 array
template <class R>
__global__ void k_pi_count(int m, const R* x0, const R* x1, const R* x2, const double* aos, int n, int* count, int* key_out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const double q[3] = {aos ? aos[3 * p] : rd_pos(x0, p), aos ? aos[3 * p + 1] : rd_pos(x1, p), aos ? aos[3 * p + 2] : rd_pos(x2, p)};
    int c[3];
    cell_of_point(n, q, c);
    const int key = vox_key(n, c[0], c[1], c[2]);
    key_out[p] = key;
    atomicAdd(count + key, 1);
}
template <class R>
__global__ void k_pi_fill(int m, const R* x0, const R* x1, const R* x2, const double* aos, const int* key, const int* cell_start,
                          int* fill, const int* orig_id, int* ids, int* slots, double* pts) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const int k = key[p];
    const int q = cell_start[k] + atomicAdd(fill + k, 1);
    ids[q] = orig_id ? orig_id[p] : p;
    slots[q] = p;
    pts[3 * q] = aos ? aos[3 * p] : rd_pos(x0, p);
    pts[3 * q + 1] = aos ? aos[3 * p + 1] : rd_pos(x1, p);
    pts[3 * q + 2] = aos ? aos[3 * p + 2] : rd_pos(x2, p);
}

struct Best { double d2; int id, slot; double p[3]; };
__device__ __forceinline__ void scan_range(const PointIndex& I, int s, int e, const double* q, Best& b) {
    for (int t = s; t < e; ++t) {
        const double dx = q[0] - I.pts[3 * t], dy = q[1] - I.pts[3 * t + 1], dz = q[2] - I.pts[3 * t + 2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        const int id = I.ids[t];
        if (d2 < b.d2 || (d2 == b.d2 && id < b.id)) {
            b.d2 = d2; b.id = id; b.slot = I.slots[t];
            b.p[0] = I.pts[3 * t]; b.p[1] = I.pts[3 * t + 1]; b.p[2] = I.pts[3 * t + 2];
        }
    }
}
__device__ __forceinline__ double box_dist2(const double* q, const double* lo, double w) {
    double s = 0;
    for (int d = 0; d < 3; ++d) {
        const double a = lo[d] - q[d], b = q[d] - (lo[d] + w);
        const double g = a > 0 ? a : (b > 0 ? b : 0);
        s += g * g;
    }
    return s;
}

__device__ __forceinline__ Best nearest(const PointIndex& I, const double* q) {
    Best b = {1e300, 0x7fffffff, -1, {0.0, 0.0, 0.0}};
    const int n = I.n;
    const double h = 1.0 / n;
    int c[3];
    cell_of_point(n, q, c);
    bool done = false;
    for (int r = 0; r <= 2 && !done; ++r) {
        for (int i = c[0] - r; i <= c[0] + r; ++i) {
            if (i < 0 || i >= n) continue;
            for (int j = c[1] - r; j <= c[1] + r; ++j) {
                if (j < 0 || j >= n) continue;
                const bool shell_ij = (i == c[0] - r || i == c[0] + r || j == c[1] - r || j == c[1] + r);
                for (int k = c[2] - r; k <= c[2] + r; k += (shell_ij || r == 0) ? 1 : 2 * r) {
                    if (k < 0 || k >= n) continue;
                    const int key = vox_key(n, i, j, k);
                    scan_range(I, I.cell_start[key], I.cell_start[key + 1], q, b);
                }
            }
        }
        // everything outside ring r is at least r*h away from q (q lies inside its own cell)
        if (r >= 1 && b.d2 <= (r * h) * (r * h)) done = true;
    }
    if (done) return b;
    // coarse level: 8^3 fine cells per box
    const int nc = n >> 3;
    const double H = 8.0 * h;
    int first = -1;
    double first_lb = 1e300;
    for (int cc = 0; cc < nc * nc * nc; ++cc) {
        if (I.cell_start[cc * 512] == I.cell_start[(cc + 1) * 512]) continue;
        const double lo[3] = {(cc / (nc * nc)) * H, ((cc / nc) % nc) * H, (cc % nc) * H};
        const double lb = box_dist2(q, lo, H);
        if (lb < first_lb) { first_lb = lb; first = cc; }
    }
    if (first >= 0) scan_range(I, I.cell_start[first * 512], I.cell_start[(first + 1) * 512], q, b);
    for (int cc = 0; cc < nc * nc * nc; ++cc) {
        if (cc == first || I.cell_start[cc * 512] == I.cell_start[(cc + 1) * 512]) continue;
        const double lo[3] = {(cc / (nc * nc)) * H, ((cc / nc) % nc) * H, (cc % nc) * H};
        if (box_dist2(q, lo, H) <= b.d2) scan_range(I, I.cell_start[cc * 512], I.cell_start[(cc + 1) * 512], q, b);
    }
    return b;
}

// direction 1: every current particle (storage slot p of frame f) -> nearest target; d/dx_p = 2 (x_p - t_nn)
template <class R>
__global__ void k_chamfer_cur_to_target(int N, const R* x0, const R* x1, const R* x2, PointIndex T, double weight, int add_grad,
                                        R* g0, R* g1, R* g2, double* loss) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    double d2 = 0;
    if (p < N) {
        const double q[3] = {rd_pos(x0, p), rd_pos(x1, p), rd_pos(x2, p)};
        const Best b = nearest(T, q);
        if (b.slot >= 0) {
            d2 = b.d2;
            if (add_grad) {
                g0[poff(p)] += (R)(2.0 * weight * (q[0] - b.p[0]));
                g1[poff(p)] += (R)(2.0 * weight * (q[1] - b.p[1]));
                g2[poff(p)] += (R)(2.0 * weight * (q[2] - b.p[2]));
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) d2 += __shfl_xor(d2, o, 64);
    if ((threadIdx.x & 63) == 0 && d2 != 0) atomicAdd(loss, d2);
}

// direction 2: every target -> nearest current particle (index C over frame f); d/dx_nn += 2 (x_nn - t_j)
template <class R>
__global__ void k_chamfer_target_to_cur(int M, const double* target, PointIndex Cidx, double weight, int add_grad, R* g0, R* g1,
                                        R* g2, double* loss) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    double d2 = 0;
    if (j < M) {
        const double q[3] = {target[3 * j], target[3 * j + 1], target[3 * j + 2]};
        const Best b = nearest(Cidx, q);
        if (b.slot >= 0) {
            d2 = b.d2;
            if (add_grad) {
                atomicAdd(g0 + poff(b.slot), (R)(2.0 * weight * (b.p[0] - q[0])));
                atomicAdd(g1 + poff(b.slot), (R)(2.0 * weight * (b.p[1] - q[1])));
                atomicAdd(g2 + poff(b.slot), (R)(2.0 * weight * (b.p[2] - q[2])));
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) d2 += __shfl_xor(d2, o, 64);
    if ((threadIdx.x & 63) == 0 && d2 != 0) atomicAdd(loss, d2);
}

// Contact-distance term of the door / transport losses (losses/loss_door.py:49-61, loss_transport.py:58-76):
//   d_i = max(|x_i - c|^2 - offset, 0) over the particles whose caller-side id lies in [id0, id1);   value = min_i d_i
// The winner is kept as (float bits of d, storage slot) packed in 64 bits, so one atomicMin finds value and argmin.
template <class R>
__global__ void k_min_dist(int N, const R* x0, const R* x1, const R* x2, const int* orig_id, int id0, int id1, double cx, double cy,
                           double cz, double offset, unsigned long long* best) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long key = ~0ull;
    if (p < N) {
        const int id = orig_id ? orig_id[p] : p;
        if (id >= id0 && id < id1) {
            const double dx = rd_pos(x0, p) - cx, dy = rd_pos(x1, p) - cy, dz = rd_pos(x2, p) - cz;
            const float d = (float)fmax(dx * dx + dy * dy + dz * dz - offset, 0.0);
            key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)p;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o, 64);
        key = other < key ? other : key;
    }
    if ((threadIdx.x & 63) == 0 && key != ~0ull && key < *best) atomicMin(best, key);
}
// value (f64, recomputed from the winner) and, on request, the seed of loss = weight * value^2 into x.grad[f][slot]
template <class R>
__global__ void k_min_dist_finish(const R* x0, const R* x1, const R* x2, const unsigned long long* best, double cx, double cy, double cz,
                                  double offset, double weight, int add_grad, R* g0, R* g1, R* g2, double* out /* value, gcx, gcy, gcz */) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    out[0] = out[1] = out[2] = out[3] = 0.0;
    const unsigned long long key = *best;
    if (key == ~0ull) return;
    const int p = (int)(key & 0xffffffffu);
    const double dx = rd_pos(x0, p) - cx, dy = rd_pos(x1, p) - cy, dz = rd_pos(x2, p) - cz;
    const double raw = dx * dx + dy * dy + dz * dz - offset;
    const double v = raw > 0.0 ? raw : 0.0;
    out[0] = v;
    if (v > 0.0) {                                   // d(weight v^2)/dx = 2 weight v * 2 (x - c);  -that for the centre
        const double k = 4.0 * weight * v;
        out[1] = -k * dx; out[2] = -k * dy; out[3] = -k * dz;
        if (add_grad) { g0[poff(p)] += (R)(k * dx); g1[poff(p)] += (R)(k * dy); g2[poff(p)] += (R)(k * dz); }
    }
}

}  // namespace smac
