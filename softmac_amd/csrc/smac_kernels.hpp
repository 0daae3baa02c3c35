// HIP kernels of the MPM substep (forward + adjoint) for gfx950.
//
// Data layout in HBM (per handle):
//   particle frames  S[f][c][p]   c = 0..23 (x3 v3 C9 E9, E = F - I), SoA so that a wave reads 64
//                                 consecutive scalars of one component (256 B / 512 B per instruction);
//   adjoint frames   A[f][c][p]   same shape;
//   grid             n^3 cells in 4x4x4 blocks, block-major (smac_sort.hpp), one array per scalar:
//                    m, v_in[3], v_mixed[3], v_out[3] and the same ten for the adjoints; only the
//                    ACTIVE blocks of the current epoch are cleared and swept;
//   particles        sorted by (block, rank-in-cell); one workgroup per chunk (<= 256 particles of
//                    one block); scatters accumulate in an 8x8x8-node LDS tile and are flushed once;
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
#include "smac_sort.hpp"

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
    const int* control_idx;      // indexed by ORIGINAL particle id
    R* action;
    R* action_grad;
    size_t G;
    // epoch data (smac_sort.hpp)
    int nb;                      // blocks per dimension
    const Chunk* chunks;
    int nchunks;
    const int* active;
    int nactive;
    const int* orig_id;          // sorted slot -> original particle id
    const R* An;                 // adjoint of frame f+1 in THIS epoch's particle order (A[f+1] or a re-ordered copy)
    R* slab;                     // [nchunks][4][TILE_WORDS] per-chunk tiles (P2G: m,vin ; G2P adjoint: agvout)
    const int* block_chunk_start;   // per block: first chunk / number of chunks (dense, nb^3)
    const int* block_chunks;
    int* drift_flag;
};

// LDS tile: the 6x6x6 nodes a particle whose base lies in a 4x4x4 block can touch (origin = 4*block).
// A chunk accumulates its scatter there with LDS atomics and stores the tile ONCE, coalesced, to its
// slab in HBM; the grid kernels then sum, per node, the <= 8 slabs that overlap it.  Particles that
// drifted out of their block since the last sort fall back to global atomics on the dense arrays.
constexpr int TW = 6, TSY = 6, TSX = 36, TILE_WORDS = 216;
__device__ __forceinline__ int tile_index(int li, int lj, int lk) { return li * TSX + lj * TSY + lk; }

// LDS tiles accumulate in f64 whatever R is: measured on gfx950 (tools/microbench/lds_atomics.hip),
// ds_add_f64 retires a conflict-free wave instruction in ~10 cycles while ds_add_f32 is lane-serial
// (~193 cycles) - and the f64 sum is the more accurate one anyway.
typedef double tile_t;
template <class R> __device__ __forceinline__ void lds_add(tile_t* p, R v) {
    __hip_atomic_fetch_add(p, (tile_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

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

// Addresses of the 27 stencil nodes: block-major global cell = cx[i] + cy[j] + cz[k]; tile-local word
// = tx[i] + ty[j] + tz[k] when the node lies inside the chunk's 6^3 tile (bit i of okx etc.).
// The base is clamped into the grid for ADDRESSING only (the reference would touch memory outside its
// fields for a particle that left the [1.5dx, 1-1.5dx] box; we must not).
struct Nodes {
    int cx[3], cy[3], cz[3];
    int tx[3], ty[3], tz[3];
    int okx, oky, okz;
    __device__ __forceinline__ int cell(int i, int j, int k) const { return cx[i] + cy[j] + cz[k]; }
    __device__ __forceinline__ bool in_tile(int i, int j, int k) const { return ((okx >> i) & (oky >> j) & (okz >> k) & 1) != 0; }
    __device__ __forceinline__ int tile(int i, int j, int k) const { return tx[i] + ty[j] + tz[k]; }
};
template <class R> __device__ __forceinline__ void stencil_at(const DevSim<R>& D, const R* x, Stencil<R>& st, Nodes& nd, int block) {
    make_stencil(x, D.inv_dx, st);
    const int nb = D.nb;
    const int bz = block % nb, by = (block / nb) % nb, bx = block / (nb * nb);
    const int org[3] = {4 * bx, 4 * by, 4 * bz};
    int cb[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int b = st.base[d];
        cb[d] = b < 0 ? 0 : (b > D.n - 3 ? D.n - 3 : b);
    }
    nd.okx = nd.oky = nd.okz = 0;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const int i = cb[0] + o, j = cb[1] + o, k = cb[2] + o;
        nd.cx[o] = ((i >> 2) * nb * nb) * 64 + ((i & 3) << 4);
        nd.cy[o] = ((j >> 2) * nb) * 64 + ((j & 3) << 2);
        nd.cz[o] = (k >> 2) * 64 + (k & 3);
        const int li = i - org[0], lj = j - org[1], lk = k - org[2];
        nd.tx[o] = li * TSX; nd.ty[o] = lj * TSY; nd.tz[o] = lk;
        nd.okx |= ((unsigned)li < (unsigned)TW) << o;
        nd.oky |= ((unsigned)lj < (unsigned)TW) << o;
        nd.okz |= ((unsigned)lk < (unsigned)TW) << o;
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
// chunk prologue shared by the particle kernels
// ------------------------------------------------------------------------------------------
#define SMAC_CHUNK_PROLOGUE                                   \
    const Chunk ch = D.chunks[blockIdx.x];                    \
    const int t = threadIdx.x;                                \
    const bool valid = t < ch.count;                          \
    const int p = ch.start + (valid ? t : 0);

// store this chunk's LDS tile (NS scalars) to its slab, coalesced
template <class R, int NS> __device__ __forceinline__ void tile_store(const DevSim<R>& D, const tile_t* tile) {
    R* dst = D.slab + (size_t)blockIdx.x * 4 * TILE_WORDS;
    for (int i = threadIdx.x; i < NS * TILE_WORDS; i += BLOCK) dst[i] = (R)tile[i];
}

// Sum, for one cell of block `b`, scalar `c` of every slab that overlaps it (own block and the
// blocks at -1 along each dimension in which the cell's local coordinate is <= 1).
template <class R, int NS>
__device__ __forceinline__ void slab_reduce(const DevSim<R>& D, int b, int l, R* acc) {
    const int nb = D.nb;
    const int bz = b % nb, by = (b / nb) % nb, bx = b / (nb * nb);
    const int lx = l >> 4, ly = (l >> 2) & 3, lz = l & 3;
    const int ex = (lx <= 1 && bx > 0) ? 1 : 0, ey = (ly <= 1 && by > 0) ? 1 : 0, ez = (lz <= 1 && bz > 0) ? 1 : 0;
    for (int dx = 0; dx <= ex; ++dx)
        for (int dy = 0; dy <= ey; ++dy)
            for (int dz = 0; dz <= ez; ++dz) {
                const int src = ((bx - dx) * nb + (by - dy)) * nb + (bz - dz);
                const int nch = D.block_chunks[src];
                if (nch == 0) continue;
                const int w = tile_index(lx + 4 * dx, ly + 4 * dy, lz + 4 * dz);
                const R* sl = D.slab + (size_t)D.block_chunk_start[src] * 4 * TILE_WORDS + w;
                for (int c = 0; c < nch; ++c, sl += 4 * TILE_WORDS)
#pragma unroll
                    for (int s = 0; s < NS; ++s) acc[s] += sl[s * TILE_WORDS];
            }
}

// zero `nfields` consecutive grid fields (starting at field0 of the 20-field block) on the active blocks
template <class R>
__global__ __launch_bounds__(BLOCK) void k_clear_active(DevSim<R> D, R* base, int nfields) {
    const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= D.nactive) return;
    const size_t cell = (size_t)D.active[a] * 64 + (threadIdx.x & 63);
    for (int f = 0; f < nfields; ++f) base[(size_t)f * D.G + cell] = R(0);
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <class R, bool STORE_F>
__global__ __launch_bounds__(BLOCK) void k_p2g(DevSim<R> D, int f) {
    __shared__ tile_t tile[4 * TILE_WORDS];
    SMAC_CHUNK_PROLOGUE
    for (int i = t; i < 4 * TILE_WORDS; i += BLOCK) tile[i] = 0.0;
    __syncthreads();
    if (valid) {
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
            int ci = D.control_idx[D.orig_id[p]];
            if (ci >= 0)
                for (int d = 0; d < 3; ++d) imp[d] = R(6e-4) * D.action[3 * ci + d] * D.dt;
        }
        Stencil<R> st;
        Nodes nd;
        stencil_at(D, x, st, nd, ch.block);
        const size_t G = D.G;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const R w = st.w[i][0] * st.w[j][1] * st.w[k][2];
                    const R d0 = (R(i) - st.fx[0]) * D.dx, d1 = (R(j) - st.fx[1]) * D.dx, d2 = (R(k) - st.fx[2]) * D.dx;
                    R mom[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        mom[c] = w * (D.p_mass * v[c] + aff[3 * c] * d0 + aff[3 * c + 1] * d1 + aff[3 * c + 2] * d2 + imp[c]);   // :261
                    if (nd.in_tile(i, j, k)) {
                        tile_t* tp = tile + nd.tile(i, j, k);
                        lds_add(tp, w * D.p_mass);                                         // :262
                        lds_add(tp + TILE_WORDS, mom[0]);
                        lds_add(tp + 2 * TILE_WORDS, mom[1]);
                        lds_add(tp + 3 * TILE_WORDS, mom[2]);
                    } else {                                                               // drifted out of the block
                        const size_t cell = nd.cell(i, j, k);
                        atomic_add(D.gm + cell, w * D.p_mass);
#pragma unroll
                        for (int c = 0; c < 3; ++c) atomic_add(D.gvin + c * G + cell, mom[c]);
                    }
                }
    }
    __syncthreads();
    tile_store<R, 4>(D, tile);
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

// one thread per cell of an active block; returns false past the end
template <class R> __device__ __forceinline__ bool active_cell(const DevSim<R>& D, int& b, int& l, size_t& cell, int& i, int& j, int& k) {
    const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= D.nactive) return false;
    b = D.active[a];
    l = threadIdx.x & 63;
    cell = (size_t)b * 64 + l;
    const int nb = D.nb;
    i = 4 * (b / (nb * nb)) + (l >> 4);
    j = 4 * ((b / nb) % nb) + ((l >> 2) & 3);
    k = 4 * (b % nb) + (l & 3);
    return true;
}

// slab reduction (completes P2G) fused with grid_op :283-297 / grid_op_mixed1 :396-404
template <class R>
__global__ __launch_bounds__(BLOCK) void k_grid_op(DevSim<R> D) {
    int b, l, i, j, k;
    size_t cell;
    if (!active_cell(D, b, l, cell, i, j, k)) return;
    R acc[4] = {D.gm[cell], D.gvin[cell], D.gvin[D.G + cell], D.gvin[2 * D.G + cell]};   // drift fallback part
    slab_reduce<R, 4>(D, b, l, acc);
    const R m = acc[0];
    D.gm[cell] = m;
#pragma unroll
    for (int d = 0; d < 3; ++d) D.gvin[d * D.G + cell] = acc[1 + d];
    if (!(m > R(1e-10))) return;                                                       // :286 / :399
    const R inv = R(1) / m;
    R v[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) v[d] = inv * acc[1 + d] + D.dt * D.g[d];               // :287-288
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
                                                              const Nodes& nd, R* out) {
    out[0] = out[1] = out[2] = R(0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const R w = st.w[i][0] * st.w[j][1] * st.w[k][2];
                const size_t cell = nd.cell(i, j, k);
#pragma unroll
                for (int c = 0; c < 3; ++c) out[c] += w * field[c * D.G + cell];
            }
}

template <class R>
__global__ __launch_bounds__(BLOCK) void k_contact(DevSim<R> D, int f) {
    SMAC_CHUNK_PROLOGUE
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
        Nodes nd;
        stencil_at(D, x, st, nd, ch.block);
        R v_tmp[3], v_tgt[3];
        gather_vec(D, D.gvmix, st, nd, v_tmp);                                          // mixed2
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
                    const size_t cell = nd.cell(i, j, k);
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
    SMAC_CHUNK_PROLOGUE
    if (!valid) return;
    const R* Sf = frame(D.S, f, D.Npad);
    R* Sn = frame(D.S, f + 1, D.Npad);
    R x[3];
    load_vec(Sf, CX, 3, D.Npad, p, x);
    Stencil<R> st;
    Nodes nd;
    stencil_at(D, x, st, nd, ch.block);
    if ((nd.okx & nd.oky & nd.okz) != 7) {          // stencil left the chunk's tile: how far has this particle drifted?
        const int nb = D.nb;
        const int pb[3] = {st.base[0] >> 2, st.base[1] >> 2, st.base[2] >> 2};
        const int cbk[3] = {ch.block / (nb * nb), (ch.block / nb) % nb, ch.block % nb};
        for (int d = 0; d < 3; ++d)
            if (pb[d] < cbk[d] - 1 || pb[d] > cbk[d] + 1) *D.drift_flag = 1;            // beyond the active halo
    }
    R nv[3] = {R(0), R(0), R(0)}, nC[9] = {R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0)};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const R w = st.w[i][0] * st.w[j][1] * st.w[k][2];
                const R dp[3] = {R(i) - st.fx[0], R(j) - st.fx[1], R(k) - st.fx[2]};
                const size_t cell = nd.cell(i, j, k);
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
    __shared__ tile_t tile[3 * TILE_WORDS];
    SMAC_CHUNK_PROLOGUE
    for (int i = t; i < 3 * TILE_WORDS; i += BLOCK) tile[i] = 0.0;
    __syncthreads();
    if (valid) {
        const R* Sf = frame(D.S, f, D.Npad);
        const R* An = D.An;
        R* Af = frame(D.A, f, D.Npad);
        R x[3], gx1[3], gv1[3], gC1[9];
        load_vec(Sf, CX, 3, D.Npad, p, x);
        load_vec(An, CX, 3, D.Npad, p, gx1);
        load_vec(An, CV, 3, D.Npad, p, gv1);
        load_vec(An, CC, 9, D.Npad, p, gC1);
        Stencil<R> st;
        Nodes nd;
        stencil_at(D, x, st, nd, ch.block);
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
                    const size_t cell = nd.cell(i, j, k);
                    const bool in = nd.in_tile(i, j, k);
                    tile_t* tp = tile + nd.tile(i, j, k);
                    R gw = R(0);
                    R gdp[3] = {R(0), R(0), R(0)};
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const R gvn = D.gvout[c * D.G + cell];
                        // d(out)/d g_v[c] = w (gnv[c] + sum_d gC[c][d] dp[d])
                        const R tt = gnv[c] + gC1[3 * c] * dp[0] + gC1[3 * c + 1] * dp[1] + gC1[3 * c + 2] * dp[2];
                        if (in) lds_add(tp + c * TILE_WORDS, w * tt);
                        else atomic_add(D.agvout + c * D.G + cell, w * tt);
                        gw += gvn * tt;
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
    __syncthreads();
    tile_store<R, 3>(D, tile);
}

// completes the G2P-adjoint scatter: agvout += sum of overlapping slabs
template <class R>
__global__ __launch_bounds__(BLOCK) void k_reduce_agvout(DevSim<R> D) {
    int b, l, i, j, k;
    size_t cell;
    if (!active_cell(D, b, l, cell, i, j, k)) return;
    R acc[3] = {D.agvout[cell], D.agvout[D.G + cell], D.agvout[2 * D.G + cell]};
    slab_reduce<R, 3>(D, b, l, acc);
#pragma unroll
    for (int d = 0; d < 3; ++d) D.agvout[d * D.G + cell] = acc[d];
}

template <class R>
__global__ __launch_bounds__(BLOCK) void k_contact_grad(DevSim<R> D, int f) {
    SMAC_CHUNK_PROLOGUE
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
        Nodes nd;
        stencil_at(D, x, st, nd, ch.block);
        const R life = R(1) / R(D.substeps - f % D.substeps);
        // recompute the forward chain, keeping the velocity entering each primitive
        R v_tmp[3], vin[MAX_PRIMS][3], vcur[3], dummy[6];
        gather_vec(D, D.gvmix, st, nd, v_tmp);
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
                    const size_t cell = nd.cell(i, j, k);
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
                    const size_t cell = nd.cell(i, j, k);
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
    int b, l, i, j, k;
    size_t cell;
    if (!active_cell(D, b, l, cell, i, j, k)) return;
    const R m = D.gm[cell];
    if (!(m > R(1e-10))) return;
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
    SMAC_CHUNK_PROLOGUE
    if (!valid) return;
    const R* Sf = frame(D.S, f, D.Npad);
    const R* An = D.An;
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
        ci = D.control_idx[D.orig_id[p]];
        if (ci >= 0)
            for (int d = 0; d < 3; ++d) imp[d] = R(6e-4) * D.action[3 * ci + d] * D.dt;
    }
    Stencil<R> st;
    Nodes nd;
    stencil_at(D, x, st, nd, ch.block);
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
                const size_t cell = nd.cell(i, j, k);
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
// compute_grid_m_kernel :607-617 - dense (i,j,k) row-major output, particle order irrelevant
template <class R>
__global__ void k_grid_m_only(const R* x0, const R* x1, const R* x2, int N, int n, R inv_dx, R p_mass, R* out) {
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= N) return;
    const R x[3] = {x0[p], x1[p], x2[p]};
    Stencil<R> st;
    make_stencil(x, inv_dx, st);
    int cb[3];
    for (int d = 0; d < 3; ++d) cb[d] = st.base[d] < 0 ? 0 : (st.base[d] > n - 3 ? n - 3 : st.base[d]);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                const size_t cell = ((size_t)(cb[0] + i) * n + (cb[1] + j)) * n + (cb[2] + k);
                atomic_add(out + cell, st.w[i][0] * st.w[j][1] * st.w[k][2] * p_mass);
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
