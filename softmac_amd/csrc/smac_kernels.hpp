// HIP kernels of the MPM substep (forward + adjoint) for gfx950.
//
// Data layout in HBM (per handle):
//   particle frames  S[f][c][p]   c = 0..23 (x3 v3 C9 E9, E = F - I), SoA so that a wave reads 64
//                                 consecutive scalars of one component (256 B / 512 B per instruction);
//   adjoint frames   A[f][c][p]   same shape;
//   grid             dense n^3, one array per scalar: m, v_in[3], v_mixed[3], v_out[3] and the
//                    same ten for the adjoints; cell = (i*n + j)*n + k;
//   primitives       state[P][max_frames][13], grad[P][max_frames][13], ext_f[P][6], ext_f_grad[P][6].
//
// Kernel <-> reference map (softmac/engine/mpm_simulator.py):
//   k_p2g            compute_F_tmp :125-128 + svd :130-133 + p2g :198-262      (fused: F_tmp,U,sig,V stay in registers)
//   k_grid_op        grid_op :283-297 / grid_op_mixed1 :396-404
//   k_contact        grid_op_mixed2 :406-419 + mixed3 :421-429 + mixed4 :431-443 (fused per particle)
//   k_g2p            g2p :299-318
//   k_g2p_grad       g2p.grad
//   k_contact_grad   grid_op_mixed4.grad + mixed3.grad + mixed2.grad
//   k_grid_op_grad   grid_op_mixed1.grad / grid_op.grad
//   k_p2g_grad       p2g.grad + svd_grad :135-157 + compute_F_tmp.grad
#pragma once
#include <hip/hip_runtime.h>
#include "smac_math.hpp"

namespace smac {

constexpr int BLOCK = 256;
enum { CX = 0, CV = 3, CC = 6, CF = 15, NCOMP = 24 };

template <class R> struct DevSim {
    int N, Npad, n, P, n_control, substeps, collision_type, sticky, max_frames;
    R dt, inv_dx, dx, p_mass, stress_scale;
    R g[3];
    Material<R> mat;
    R* S;
    R* A;
    R *gm, *gvin, *gvmix, *gvout;       // grid values  (vectors: [3][G])
    R *agm, *agvin, *agvmix, *agvout;   // grid adjoints
    PrimTable<R> prim[MAX_PRIMS];
    R* prim_state;
    R* prim_grad;
    R* ext_f;
    R* ext_f_grad;
    const int* control_idx;
    R* action;
    R* action_grad;
    size_t G;
};

template <class R> __device__ __forceinline__ void atomic_add(R* p, R v) { unsafeAtomicAdd(p, v); }

template <class R> __device__ __forceinline__ R wave_sum(R v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

template <class R> __device__ __forceinline__ const R* frame(const R* base, int f, int Npad) {
    return base + (size_t)f * NCOMP * Npad;
}
template <class R> __device__ __forceinline__ R* frame(R* base, int f, int Npad) {
    return base + (size_t)f * NCOMP * Npad;
}

template <class R> __device__ __forceinline__ void load_vec(const R* fr, int c0, int cnt, int Npad, int p, R* out) {
#pragma unroll
    for (int i = 0; i < cnt; ++i) out[i] = fr[(size_t)(c0 + i) * Npad + p];
}

// Stencil with the base clamped into the grid for ADDRESSING only (the reference would touch
// memory outside its fields for a particle that left the [1.5dx, 1-1.5dx] box; we must not).
template <class R> __device__ __forceinline__ void stencil_at(const DevSim<R>& D, const R* x, Stencil<R>& st, int* cb) {
    make_stencil(x, D.inv_dx, st);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int b = st.base[d];
        b = b < 0 ? 0 : (b > D.n - 3 ? D.n - 3 : b);
        cb[d] = b;
    }
}

// F_tmp - I = E + dt (C + C E)     (compute_F_tmp, :125-128)
template <class R> __device__ __forceinline__ void f_tmp(const R* C, const R* E, R dt, R* Et) {
    R CE[9];
    mm(C, E, CE);
#pragma unroll
    for (int i = 0; i < 9; ++i) Et[i] = E[i] + dt * (C[i] + CE[i]);
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <class R, bool STORE_F>
__global__ __launch_bounds__(BLOCK) void k_p2g(DevSim<R> D, int f) {
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= D.N) return;
    const R* Sf = frame(D.S, f, D.Npad);
    R x[3], v[3], C[9], E[9], Et[9], En[9], stress[9], aff[9];
    load_vec(Sf, CX, 3, D.Npad, p, x);
    load_vec(Sf, CV, 3, D.Npad, p, v);
    load_vec(Sf, CC, 9, D.Npad, p, C);
    load_vec(Sf, CF, 9, D.Npad, p, E);
    f_tmp(C, E, D.dt, Et);
    ConstState<R> cs;
    constitutive_fwd(D.mat, Et, En, stress, cs);
    if (STORE_F) {
        R* Sn = frame(D.S, f + 1, D.Npad);
#pragma unroll
        for (int i = 0; i < 9; ++i) Sn[(size_t)(CF + i) * D.Npad + p] = En[i];     // :250
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) aff[i] = D.stress_scale * stress[i] + D.p_mass * C[i];   // :247-248
    R imp[3] = {R(0), R(0), R(0)};
    if (D.n_control > 0) {                                                            // :209-213
        int ci = D.control_idx[p];
        if (ci >= 0)
            for (int d = 0; d < 3; ++d) imp[d] = R(6e-4) * D.action[3 * ci + d] * D.dt;
    }
    Stencil<R> st;
    int cb[3];
    stencil_at(D, x, st, cb);
    const size_t G = D.G;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const R w = st.w[i][0] * st.w[j][1] * st.w[k][2];
                const R d0 = (R(i) - st.fx[0]) * D.dx, d1 = (R(j) - st.fx[1]) * D.dx, d2 = (R(k) - st.fx[2]) * D.dx;
                const size_t cell = ((size_t)(cb[0] + i) * D.n + (cb[1] + j)) * D.n + (cb[2] + k);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    R mom = D.p_mass * v[c] + aff[3 * c] * d0 + aff[3 * c + 1] * d1 + aff[3 * c + 2] * d2 + imp[c];
                    atomic_add(D.gvin + c * G + cell, w * mom);                         // :261
                }
                atomic_add(D.gm + cell, w * D.p_mass);                                  // :262
            }
}

// boundary_condition :268-281 on a velocity; returns mask bits of the components that were zeroed
template <class R> __device__ __forceinline__ int boundary(const DevSim<R>& D, int i, int j, int k, R* v) {
    const int I[3] = {i, j, k};
    int mask = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (I[d] < 3 && v[d] < R(0)) { v[d] = R(0); mask |= 1 << d; }
        if (I[d] > D.n - 3 && v[d] > R(0)) { v[d] = R(0); mask |= 1 << d; }
    }
    if (D.sticky && j < 3) { v[0] = v[1] = v[2] = R(0); mask = 7; }                   // :278-279
    return mask;
}

template <class R>
__global__ __launch_bounds__(BLOCK) void k_grid_op(DevSim<R> D) {
    const size_t cell = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (cell >= D.G) return;
    const R m = D.gm[cell];
    if (!(m > R(1e-10))) return;                                                       // :286 / :399
    const int k = cell % D.n, j = (cell / D.n) % D.n, i = cell / ((size_t)D.n * D.n);
    const R inv = R(1) / m;
    R v[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) v[d] = inv * D.gvin[d * D.G + cell] + D.dt * D.g[d];   // :287-288
    boundary(D, i, j, k, v);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (D.collision_type == CONTACT_MIXED) D.gvmix[d * D.G + cell] = v[d];          // :403
        D.gvout[d * D.G + cell] = v[d];                                                 // :404 / :297
    }
}

// band test shared by k_contact and k_contact_grad: which primitives see this particle
template <class R> __device__ __forceinline__ int contact_mask(const DevSim<R>& D, int f, const R* x) {
    int mask = 0;
#pragma unroll
    for (int i = 0; i < MAX_PRIMS; ++i) {
        if (i >= D.P || !D.prim[i].contact) continue;
        const R* st = D.prim_state + ((size_t)i * D.max_frames + f) * 13;
        R d = prim_sdf(D.prim[i], st, x);
        if (d <= R(5e-3)) mask |= 1 << i;
    }
    return mask;
}

template <class R> __device__ __forceinline__ void gather_vec(const DevSim<R>& D, const R* field, const Stencil<R>& st,
                                                              const int* cb, R* out) {
    out[0] = out[1] = out[2] = R(0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const R w = st.w[i][0] * st.w[j][1] * st.w[k][2];
                const size_t cell = ((size_t)(cb[0] + i) * D.n + (cb[1] + j)) * D.n + (cb[2] + k);
#pragma unroll
                for (int c = 0; c < 3; ++c) out[c] += w * field[c * D.G + cell];
            }
}

template <class R>
__global__ __launch_bounds__(BLOCK) void k_contact(DevSim<R> D, int f) {
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = p < D.N;
    R x[3] = {R(0.5), R(0.5), R(0.5)};
    if (valid) load_vec(frame(D.S, f, D.Npad), CX, 3, D.Npad, p, x);
    const int mask = valid ? contact_mask(D, f, x) : 0;
    R ext[MAX_PRIMS][6];
#pragma unroll
    for (int i = 0; i < MAX_PRIMS; ++i)
#pragma unroll
        for (int c = 0; c < 6; ++c) ext[i][c] = R(0);
    if (mask) {
        Stencil<R> st;
        int cb[3];
        stencil_at(D, x, st, cb);
        R v_tmp[3], v_tgt[3];
        gather_vec(D, D.gvmix, st, cb, v_tmp);                                          // mixed2
        v_tgt[0] = v_tmp[0]; v_tgt[1] = v_tmp[1]; v_tgt[2] = v_tmp[2];
        const R life = R(1) / R(D.substeps - f % D.substeps);                           // :425
#pragma unroll
        for (int i = 0; i < MAX_PRIMS; ++i)                                             // mixed3
            if (mask & (1 << i)) {
                const R* ps = D.prim_state + ((size_t)i * D.max_frames + f) * 13;
                R s13[13];
                for (int c = 0; c < 13; ++c) s13[c] = ps[c];
                collide_mixed(D.prim[i], s13, x, v_tgt, D.p_mass, D.dt, life, ext[i]);
            }
        const R diff[3] = {v_tmp[0] - v_tgt[0], v_tmp[1] - v_tgt[1], v_tmp[2] - v_tgt[2]};
#pragma unroll
        for (int i = 0; i < 3; ++i)                                                     // mixed4
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const size_t cell = ((size_t)(cb[0] + i) * D.n + (cb[1] + j)) * D.n + (cb[2] + k);
                    if (D.gm[cell] > R(1e-10)) {
                        const R w = R(2) * st.w[i][0] * st.w[j][1] * st.w[k][2];        // alpha = 2, :437
#pragma unroll
                        for (int c = 0; c < 3; ++c) atomic_add(D.gvout + c * D.G + cell, -w * diff[c]);
                    }
                }
    }
    // ext_f: one atomic per wave per component instead of one per contacting particle
    const unsigned long long any = __ballot(mask != 0);
    if (any) {
#pragma unroll
        for (int i = 0; i < MAX_PRIMS; ++i) {
            if (i >= D.P || !__ballot(mask & (1 << i))) continue;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                R s = wave_sum(ext[i][c]);
                if ((threadIdx.x & 63) == 0) atomic_add(D.ext_f + i * 6 + c, s);
            }
        }
    }
}

template <class R>
__global__ __launch_bounds__(BLOCK) void k_g2p(DevSim<R> D, int f) {
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= D.N) return;
    const R* Sf = frame(D.S, f, D.Npad);
    R* Sn = frame(D.S, f + 1, D.Npad);
    R x[3];
    load_vec(Sf, CX, 3, D.Npad, p, x);
    Stencil<R> st;
    int cb[3];
    stencil_at(D, x, st, cb);
    R nv[3] = {R(0), R(0), R(0)}, nC[9] = {R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0)};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const R w = st.w[i][0] * st.w[j][1] * st.w[k][2];
                const R dp[3] = {R(i) - st.fx[0], R(j) - st.fx[1], R(k) - st.fx[2]};
                const size_t cell = ((size_t)(cb[0] + i) * D.n + (cb[1] + j)) * D.n + (cb[2] + k);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const R gv = w * D.gvout[c * D.G + cell];
                    nv[c] += gv;
#pragma unroll
                    for (int d = 0; d < 3; ++d) nC[3 * c + d] += gv * dp[d];
                }
            }
    const R four_inv_dx = R(4) * D.inv_dx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Sn[(size_t)(CV + c) * D.Npad + p] = nv[c];
        Sn[(size_t)(CX + c) * D.Npad + p] = x[c] + D.dt * nv[c];                       // :318
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) Sn[(size_t)(CC + c) * D.Npad + p] = four_inv_dx * nC[c];
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
// adjoint of the three per-dimension weight factors -> adjoint of fx.  gw[i][j][k] folded on the fly:
template <class R> struct WGrad {
    R g[3][3];   // g[k][d]: adjoint of st.w[k][d]
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = R(0);
    }
    __device__ __forceinline__ void add(const Stencil<R>& st, int i, int j, int k, R gw) {
        g[i][0] += gw * st.w[j][1] * st.w[k][2];
        g[j][1] += gw * st.w[i][0] * st.w[k][2];
        g[k][2] += gw * st.w[i][0] * st.w[j][1];
    }
    __device__ __forceinline__ void to_fx(const Stencil<R>& st, R* gfx) const {
#pragma unroll
        for (int d = 0; d < 3; ++d) gfx[d] += g[0][d] * st.dw[0][d] + g[1][d] * st.dw[1][d] + g[2][d] * st.dw[2][d];
    }
};

template <class R>
__global__ __launch_bounds__(BLOCK) void k_g2p_grad(DevSim<R> D, int f) {
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= D.N) return;
    const R* Sf = frame(D.S, f, D.Npad);
    const R* An = frame(D.A, f + 1, D.Npad);
    R* Af = frame(D.A, f, D.Npad);
    R x[3], gx1[3], gv1[3], gC1[9];
    load_vec(Sf, CX, 3, D.Npad, p, x);
    load_vec(An, CX, 3, D.Npad, p, gx1);
    load_vec(An, CV, 3, D.Npad, p, gv1);
    load_vec(An, CC, 9, D.Npad, p, gC1);
    Stencil<R> st;
    int cb[3];
    stencil_at(D, x, st, cb);
    const R four_inv_dx = R(4) * D.inv_dx;
    R gnv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) gnv[c] = gv1[c] + D.dt * gx1[c];                        // x' = x + dt v'
#pragma unroll
    for (int c = 0; c < 9; ++c) gC1[c] *= four_inv_dx;
    WGrad<R> wg;
    wg.zero();
    R gfx[3] = {R(0), R(0), R(0)};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const R w = st.w[i][0] * st.w[j][1] * st.w[k][2];
                const R dp[3] = {R(i) - st.fx[0], R(j) - st.fx[1], R(k) - st.fx[2]};
                const size_t cell = ((size_t)(cb[0] + i) * D.n + (cb[1] + j)) * D.n + (cb[2] + k);
                R gw = R(0);
                R gdp[3] = {R(0), R(0), R(0)};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const R gvn = D.gvout[c * D.G + cell];
                    // d(out)/d g_v[c] = w (gnv[c] + sum_d gC[c][d] dp[d])
                    const R t = gnv[c] + gC1[3 * c] * dp[0] + gC1[3 * c + 1] * dp[1] + gC1[3 * c + 2] * dp[2];
                    atomic_add(D.agvout + c * D.G + cell, w * t);
                    gw += gvn * t;
#pragma unroll
                    for (int d = 0; d < 3; ++d) gdp[d] += gvn * gC1[3 * c + d];
                }
                wg.add(st, i, j, k, gw);
#pragma unroll
                for (int d = 0; d < 3; ++d) gfx[d] -= w * gdp[d];                        // dpos = offset - fx
            }
    wg.to_fx(st, gfx);
#pragma unroll
    for (int d = 0; d < 3; ++d) Af[(size_t)(CX + d) * D.Npad + p] += gx1[d] + D.inv_dx * gfx[d];
}

template <class R>
__global__ __launch_bounds__(BLOCK) void k_contact_grad(DevSim<R> D, int f) {
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = p < D.N;
    R x[3] = {R(0.5), R(0.5), R(0.5)};
    if (valid) load_vec(frame(D.S, f, D.Npad), CX, 3, D.Npad, p, x);
    const int mask = valid ? contact_mask(D, f, x) : 0;
    R gst[MAX_PRIMS][13];
#pragma unroll
    for (int i = 0; i < MAX_PRIMS; ++i)
#pragma unroll
        for (int c = 0; c < 13; ++c) gst[i][c] = R(0);
    if (mask) {
        Stencil<R> st;
        int cb[3];
        stencil_at(D, x, st, cb);
        const R life = R(1) / R(D.substeps - f % D.substeps);
        // recompute the forward chain, keeping the velocity entering each primitive
        R v_tmp[3], vin[MAX_PRIMS][3], vcur[3], dummy[6];
        gather_vec(D, D.gvmix, st, cb, v_tmp);
        vcur[0] = v_tmp[0]; vcur[1] = v_tmp[1]; vcur[2] = v_tmp[2];
#pragma unroll
        for (int i = 0; i < MAX_PRIMS; ++i) {
            vin[i][0] = vcur[0]; vin[i][1] = vcur[1]; vin[i][2] = vcur[2];
            if (mask & (1 << i)) {
                const R* ps = D.prim_state + ((size_t)i * D.max_frames + f) * 13;
                R s13[13];
                for (int c = 0; c < 13; ++c) s13[c] = ps[c];
                collide_mixed(D.prim[i], s13, x, vcur, D.p_mass, D.dt, life, dummy);
            }
        }
        const R diff[3] = {v_tmp[0] - vcur[0], v_tmp[1] - vcur[1], v_tmp[2] - vcur[2]};
        // mixed4.grad: gd = d/d(v_tmp - v_tgt), weight adjoints
        WGrad<R> wg;
        wg.zero();
        R gd[3] = {R(0), R(0), R(0)};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const size_t cell = ((size_t)(cb[0] + i) * D.n + (cb[1] + j)) * D.n + (cb[2] + k);
                    if (D.gm[cell] > R(1e-10)) {
                        const R w = st.w[i][0] * st.w[j][1] * st.w[k][2];
                        R dg = R(0);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const R G = D.agvout[c * D.G + cell];
                            gd[c] -= R(2) * w * G;
                            dg += diff[c] * G;
                        }
                        wg.add(st, i, j, k, -R(2) * dg);
                    }
                }
        R gpos[3] = {R(0), R(0), R(0)};
        R g[3] = {-gd[0], -gd[1], -gd[2]};             // adjoint of v_tgt
        // mixed3.grad: reverse the primitive chain
#pragma unroll
        for (int i = MAX_PRIMS - 1; i >= 0; --i)
            if (mask & (1 << i)) {
                const R* ps = D.prim_state + ((size_t)i * D.max_frames + f) * 13;
                R s13[13], ge[6], gin[3];
                for (int c = 0; c < 13; ++c) s13[c] = ps[c];
                for (int c = 0; c < 6; ++c) ge[c] = D.ext_f_grad[i * 6 + c];
                collide_mixed_adjoint(D.prim[i], s13, x, vin[i], D.p_mass, D.dt, life, g, ge, gpos, gin, gst[i]);
                g[0] = gin[0]; g[1] = gin[1]; g[2] = gin[2];
            }
        // adjoint of v_tmp = direct (mixed4) + through the chain (mixed3)
        const R gvt[3] = {gd[0] + g[0], gd[1] + g[1], gd[2] + g[2]};
        // mixed2.grad
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const R w = st.w[i][0] * st.w[j][1] * st.w[k][2];
                    const size_t cell = ((size_t)(cb[0] + i) * D.n + (cb[1] + j)) * D.n + (cb[2] + k);
                    R gw = R(0);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        atomic_add(D.agvmix + c * D.G + cell, w * gvt[c]);
                        gw += D.gvmix[c * D.G + cell] * gvt[c];
                    }
                    wg.add(st, i, j, k, gw);
                }
        R gfx[3] = {R(0), R(0), R(0)};
        wg.to_fx(st, gfx);
        R* Af = frame(D.A, f, D.Npad);
#pragma unroll
        for (int d = 0; d < 3; ++d) Af[(size_t)(CX + d) * D.Npad + p] += gpos[d] + D.inv_dx * gfx[d];
    }
    const unsigned long long any = __ballot(mask != 0);
    if (any) {
#pragma unroll
        for (int i = 0; i < MAX_PRIMS; ++i) {
            if (i >= D.P || !__ballot(mask & (1 << i))) continue;
            R* pg = D.prim_grad + ((size_t)i * D.max_frames + f) * 13;
#pragma unroll
            for (int c = 0; c < 13; ++c) {
                R s = wave_sum(gst[i][c]);
                if ((threadIdx.x & 63) == 0) atomic_add(pg + c, s);
            }
        }
    }
}

template <class R>
__global__ __launch_bounds__(BLOCK) void k_grid_op_grad(DevSim<R> D) {
    const size_t cell = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (cell >= D.G) return;
    const R m = D.gm[cell];
    if (!(m > R(1e-10))) return;
    const int k = cell % D.n, j = (cell / D.n) % D.n, i = cell / ((size_t)D.n * D.n);
    const R inv = R(1) / m;
    R v[3], vin[3], g[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        vin[d] = D.gvin[d * D.G + cell];
        v[d] = inv * vin[d] + D.dt * D.g[d];
        g[d] = D.agvout[d * D.G + cell];
        if (D.collision_type == CONTACT_MIXED) g[d] += D.agvmix[d * D.G + cell];       // grid_v_out += grid_v_mixed
    }
    const int mask = boundary(D, i, j, k, v);
    R gm = R(0);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (mask & (1 << d)) g[d] = R(0);
        D.agvin[d * D.G + cell] = g[d] * inv;
        gm -= vin[d] * g[d];
    }
    D.agm[cell] = gm * inv * inv;
}

template <class R>
__global__ __launch_bounds__(BLOCK) void k_p2g_grad(DevSim<R> D, int f) {
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= D.N) return;
    const R* Sf = frame(D.S, f, D.Npad);
    const R* An = frame(D.A, f + 1, D.Npad);
    R* Af = frame(D.A, f, D.Npad);
    R x[3], v[3], C[9], E[9], Et[9], En[9], stress[9], aff[9];
    load_vec(Sf, CX, 3, D.Npad, p, x);
    load_vec(Sf, CV, 3, D.Npad, p, v);
    load_vec(Sf, CC, 9, D.Npad, p, C);
    load_vec(Sf, CF, 9, D.Npad, p, E);
    f_tmp(C, E, D.dt, Et);
    ConstState<R> cs;
    constitutive_fwd(D.mat, Et, En, stress, cs);
#pragma unroll
    for (int i = 0; i < 9; ++i) aff[i] = D.stress_scale * stress[i] + D.p_mass * C[i];
    R imp[3] = {R(0), R(0), R(0)};
    int ci = -1;
    if (D.n_control > 0) {
        ci = D.control_idx[p];
        if (ci >= 0)
            for (int d = 0; d < 3; ++d) imp[d] = R(6e-4) * D.action[3 * ci + d] * D.dt;
    }
    Stencil<R> st;
    int cb[3];
    stencil_at(D, x, st, cb);
    WGrad<R> wg;
    wg.zero();
    R gvp[3] = {R(0), R(0), R(0)}, gfx[3] = {R(0), R(0), R(0)};
    R gaff[9] = {R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0)};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const R w = st.w[i][0] * st.w[j][1] * st.w[k][2];
                const R dp[3] = {(R(i) - st.fx[0]) * D.dx, (R(j) - st.fx[1]) * D.dx, (R(k) - st.fx[2]) * D.dx};
                const size_t cell = ((size_t)(cb[0] + i) * D.n + (cb[1] + j)) * D.n + (cb[2] + k);
                R gw = D.agm[cell] * D.p_mass;
                R gdp[3] = {R(0), R(0), R(0)};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const R gv = D.agvin[c * D.G + cell];
                    const R mom = D.p_mass * v[c] + aff[3 * c] * dp[0] + aff[3 * c + 1] * dp[1] + aff[3 * c + 2] * dp[2] + imp[c];
                    gw += gv * mom;
                    gvp[c] += w * gv;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        gaff[3 * c + d] += w * gv * dp[d];
                        gdp[d] += aff[3 * c + d] * gv;
                    }
                }
                wg.add(st, i, j, k, gw);
#pragma unroll
                for (int d = 0; d < 3; ++d) gfx[d] -= w * gdp[d] * D.dx;                 // dpos = (offset - fx) dx
            }
    wg.to_fx(st, gfx);
    // impulse adjoint = sum_nodes w gv = gvp  -> action.grad
    if (ci >= 0)
        for (int d = 0; d < 3; ++d) atomic_add(D.action_grad + 3 * ci + d, R(6e-4) * D.dt * gvp[d]);
    // constitutive adjoint
    R G[9], gFn[9], gEt[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) G[i] = D.stress_scale * gaff[i];
    load_vec(An, CF, 9, D.Npad, p, gFn);
    constitutive_bwd(D.mat, Et, cs, G, gFn, gEt);
    // compute_F_tmp.grad: F_tmp = (I + dt C)(I + E)
    R gC[9], gE[9], Ft[9], A1[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { Ft[i] = E[i]; A1[i] = D.dt * C[i]; }
    Ft[0] += R(1); Ft[4] += R(1); Ft[8] += R(1);
    A1[0] += R(1); A1[4] += R(1); A1[8] += R(1);
    mmt(gEt, Ft, gC);          // gEt (I+E)^T
    mtm(A1, gEt, gE);          // (I + dt C)^T gEt
#pragma unroll
    for (int i = 0; i < 9; ++i) gC[i] = D.dt * gC[i] + D.p_mass * gaff[i];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        Af[(size_t)(CX + d) * D.Npad + p] += D.inv_dx * gfx[d];
        Af[(size_t)(CV + d) * D.Npad + p] += D.p_mass * gvp[d];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        Af[(size_t)(CC + i) * D.Npad + p] += gC[i];
        Af[(size_t)(CF + i) * D.Npad + p] += gE[i];
    }
}

// ------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------
template <class R>
__global__ void k_grid_m_only(DevSim<R> D, int f) {                                    // compute_grid_m_kernel :607-617
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= D.N) return;
    R x[3];
    load_vec(frame(D.S, f, D.Npad), CX, 3, D.Npad, p, x);
    Stencil<R> st;
    int cb[3];
    stencil_at(D, x, st, cb);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                const size_t cell = ((size_t)(cb[0] + i) * D.n + (cb[1] + j)) * D.n + (cb[2] + k);
                atomic_add(D.gm + cell, st.w[i][0] * st.w[j][1] * st.w[k][2] * D.p_mass);
            }
}

template <class R>
__global__ void k_count_active(const R* gm, size_t G, unsigned long long* out) {
    size_t cell = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    int a = (cell < G && gm[cell] > R(0)) ? 1 : 0;
    unsigned long long b = __ballot(a);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, (unsigned long long)__popcll(b));
}

// forward_kinematics :280-283 and its adjoint (13 inputs -> 7 outputs, forward-mode duals)
template <class R>
__global__ void k_prim_fk(R* state, int f, R dt) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    R* s = state + (size_t)f * 13;
    R o[7];
    forward_kinematics(s, dt, o);
    for (int i = 0; i < 7; ++i) s[13 + i] = o[i];
}
template <class R>
__global__ void k_prim_fk_grad(const R* state, R* grad, int f, R dt) {
    const int dir = threadIdx.x;
    if (dir >= 13 || blockIdx.x != 0) return;
    const R* s = state + (size_t)f * 13;
    Dual<R> sd[13], o[7];
    for (int i = 0; i < 13; ++i) sd[i] = Dual<R>(s[i], i == dir ? R(1) : R(0));
    forward_kinematics(sd, dt, o);
    R acc = R(0);
    for (int i = 0; i < 7; ++i) acc += grad[(size_t)(f + 1) * 13 + i] * o[i].d;
    grad[(size_t)f * 13 + dir] += acc;
}

}  // namespace smac
