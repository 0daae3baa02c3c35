// Kernels of the soft <-> cloth contact that are not part of the substep itself: contact-face search, penetration tracing, hit list.
// (The contact model runs inside k_contact_hits / k_contact_grad, CLOTH instantiation.)  Integer outputs, no gradient
// (soft_cloth/engine/mpm_simulator.py:463-469, 512-518, 547-553 replace the adjoints by no-ops).
#pragma once
#include "smac_kernels.hpp"

namespace smac {

constexpr int CLOTH_FACE_BATCH = 256;      // faces staged in LDS per pass: 256 x 9 doubles = 18 KB

// particle position of slot p in physical units (the particle kernels work on the unit domain; the cloth lives in [0, mpm_scale))
template <class R> __device__ __forceinline__ void cloth_particle_pos(const DevSim<R>& D, const R* Sf, int p, double* x) {
    typename pos_of<R>::type xp[3];
    load_pos(Sf, D.Npad, p, xp);
    for (int c = 0; c < 3; ++c) x[c] = pos_get(xp[c]) * D.cloth.par.scale;
}

// get_contact_pair_kernel :447-461: per particle the face with the smallest distance among the faces whose padded bounding box
// holds it (every face for a particle that was penetrated at the previous frame); the first minimum wins, -1 without candidate.
// One thread per particle slot; the faces' vertices are staged through LDS in batches, in face order.
template <class R>
__global__ __launch_bounds__(BLOCK) void k_cloth_pairs(DevSim<R> D, int f, const R* Sf, const int* orig) {
    __shared__ double fv[CLOTH_FACE_BATCH][9];
    const ClothDev& Cl = D.cloth;
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = p < D.N;
    double px[3] = {0, 0, 0};
    int id = 0, pen = 0;
    if (valid) {
        cloth_particle_pos(D, Sf, p, px);
        id = orig ? orig[p] : p;
        if (f > 0) pen = Cl.penetration[(size_t)(f - 1) * Cl.n_ids + id];
    }
    const double threshold = 1e-2 * Cl.par.scale;
    double dmin = 1e10;
    int best = -1;
    const double* vpos = Cl.pos + (size_t)f * Cl.V * 3;
    for (int base = 0; base < Cl.Fc; base += CLOTH_FACE_BATCH) {
        const int nb = min(CLOTH_FACE_BATCH, Cl.Fc - base);
        __syncthreads();
        for (int i = threadIdx.x; i < nb * 9; i += BLOCK) {
            const int q = i / 9, k = i % 9;
            fv[q][k] = vpos[(size_t)Cl.faces[3 * (base + q) + k / 3] * 3 + k % 3];
        }
        __syncthreads();
        if (valid)
            for (int q = 0; q < nb; ++q)
                if (pen || cl_in_bbox(px, fv[q], fv[q] + 3, fv[q] + 6, threshold)) {
                    const double d = cl_distance(px, fv[q], fv[q] + 3, fv[q] + 6);
                    if (d < dmin) { dmin = d; best = base + q; }
                }
    }
    if (valid) Cl.contact_id[(size_t)f * Cl.n_ids + id] = best;
}

// ---- broad phase (round 3): a uniform grid over the sheet's padded face boxes, at the granularity the particles are binned at (4^3-cell grid
// blocks).  face_blocks(): the blocks whose particles can lie inside the face's padded box.  A particle of a chunk of block b had its stencil base
// in the block's cells when it was binned (x in [(4b + 0.5) dx, (4b + 4.5) dx)) and may have drifted by one block's width since (beyond that the
// drift flag stops the epoch), so block b must list every face whose padded box reaches [(4b - 3.5) dx, (4b + 8.5) dx) in all three dimensions.
// Built per sheet frame by two passes over the faces (count, scan, fill); the lists are unordered, the query breaks distance ties by face id.
struct FaceHash { int* count; int* start; int* list; int cap; int* overflow; };
__device__ __forceinline__ void face_blocks(const ClothDev& Cl, const double* vpos, int q, int n, int nb, int* lo3, int* hi3) {
    const double threshold = 1e-2 * Cl.par.scale, inv = (double)n / (4.0 * Cl.par.scale);     // physical length -> block units
    for (int c = 0; c < 3; ++c) {
        const double a = vpos[(size_t)Cl.faces[3 * q] * 3 + c], b = vpos[(size_t)Cl.faces[3 * q + 1] * 3 + c], d = vpos[(size_t)Cl.faces[3 * q + 2] * 3 + c];
        const double flo = (min_(a, min_(b, d)) - threshold) * inv, fhi = (max_(a, max_(b, d)) + threshold) * inv;
        int l = (int)floor(flo - 8.5 / 4.0) + 1, h = (int)ceil(fhi + 3.5 / 4.0) - 1;
        lo3[c] = l < 0 ? 0 : l;
        hi3[c] = h > nb - 1 ? nb - 1 : h;
    }
}
template <bool FILL>
__global__ void k_cloth_hash(ClothDev Cl, int f, int n, int nb, FaceHash H) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Cl.Fc) return;
    int lo[3], hi[3];
    face_blocks(Cl, Cl.pos + (size_t)f * Cl.V * 3, q, n, nb, lo, hi);
    for (int i = lo[0]; i <= hi[0]; ++i)
        for (int j = lo[1]; j <= hi[1]; ++j)
            for (int k = lo[2]; k <= hi[2]; ++k) {
                const int b = (i * nb + j) * nb + k;
                const int slot = atomicAdd(H.count + b, 1);
                if (FILL) {
                    const int at = H.start[b] + slot;
                    if (at < H.cap) H.list[at] = q;
                    else *H.overflow = 1;                  // the list does not fit: the query falls back to the full scan (and the host grows the list)
                }
            }
}

// The same search for a frame in sorted order, one workgroup per chunk: the chunk's particles share a grid block, so the faces are
// culled once per workgroup against the chunk's bounding box (taken from the particles themselves, drift included) and only the
// survivors - compacted in face order, so the first-minimum rule is unchanged - reach the per-particle test.  A thin sheet leaves
// most chunks without a single candidate.  A chunk that holds a particle flagged as penetrated keeps every face (:457).
// HASH: the faces come from the block's list of the broad phase instead of the whole mesh (bit-equal result: a superset of the faces that pass
// the chunk cull, ties between equal distances go to the lower face id = "first minimum wins" of the in-order scan).
template <class R, bool HASH>
__global__ __launch_bounds__(BLOCK) void k_cloth_pairs_chunk(DevSim<R> D, int f, const R* Sf, FaceHash H) {
    __shared__ double fv[CLOTH_FACE_BATCH][9];
    __shared__ int fid[CLOTH_FACE_BATCH];
    __shared__ double red[2][4][3];
    __shared__ int wave_count[4], any_pen;
    SMAC_CHUNK_PROLOGUE
    const ClothDev& Cl = D.cloth;
    const int lane = t & 63, wave = t >> 6;
    double px[3] = {0, 0, 0};
    int id = 0, pen = 0;
    if (valid) {
        cloth_particle_pos(D, Sf, p, px);
        id = D.orig_id[p];
        if (f > 0) pen = Cl.penetration[(size_t)(f - 1) * Cl.n_ids + id];
    }
    if (t == 0) any_pen = 0;
    // bounding box of the chunk's particles
    double lo[3], hi[3];
    for (int c = 0; c < 3; ++c) {
        lo[c] = valid ? px[c] : 1e300; hi[c] = valid ? px[c] : -1e300;
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fmin(lo[c], __shfl_xor(lo[c], o, 64)); hi[c] = fmax(hi[c], __shfl_xor(hi[c], o, 64)); }
        if (lane == 0) { red[0][wave][c] = lo[c]; red[1][wave][c] = hi[c]; }
    }
    __syncthreads();
    if (pen) any_pen = 1;
    for (int c = 0; c < 3; ++c) {
        lo[c] = fmin(fmin(red[0][0][c], red[0][1][c]), fmin(red[0][2][c], red[0][3][c]));
        hi[c] = fmax(fmax(red[1][0][c], red[1][1][c]), fmax(red[1][2][c], red[1][3][c]));
    }
    __syncthreads();
    const bool keep_all = any_pen != 0;
    const bool listed = HASH && !keep_all && *H.overflow == 0;        // (a chunk with a penetrated particle keeps every face, :457)
    const int nfaces = listed ? H.count[ch.block] : Cl.Fc;
    const int* flist = listed ? H.list + H.start[ch.block] : nullptr;
    const double threshold = 1e-2 * Cl.par.scale;
    double dmin = 1e10;
    int best = -1;
    const double* vpos = Cl.pos + (size_t)f * Cl.V * 3;
    for (int base = 0; base < nfaces; base += CLOTH_FACE_BATCH) {
        const int q = base + t < nfaces ? (listed ? flist[base + t] : base + t) : Cl.Fc;
        double v9[9];
        bool cand = false;
        if (q < Cl.Fc) {
            for (int k = 0; k < 9; ++k) v9[k] = vpos[(size_t)Cl.faces[3 * q + k / 3] * 3 + k % 3];
            cand = true;
            if (!keep_all)
                for (int c = 0; c < 3; ++c) {
                    const double flo = min_(v9[c], min_(v9[3 + c], v9[6 + c])) - threshold, fhi = max_(v9[c], max_(v9[3 + c], v9[6 + c])) + threshold;
                    if (hi[c] <= flo || lo[c] >= fhi) cand = false;
                }
        }
        const unsigned long long m = __ballot(cand);
        if (lane == 0) wave_count[wave] = __popcll(m);
        __syncthreads();
        int slot = __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) slot += wave_count[w];
        const int ncand = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        if (cand) {
            for (int k = 0; k < 9; ++k) fv[slot][k] = v9[k];
            fid[slot] = q;
        }
        __syncthreads();
        if (valid)
            for (int j = 0; j < ncand; ++j)
                if (pen || cl_in_bbox(px, fv[j], fv[j] + 3, fv[j] + 6, threshold)) {
                    const double d = cl_distance(px, fv[j], fv[j] + 3, fv[j] + 6);
                    if (d < dmin || (HASH && d == dmin && fid[j] < best)) { dmin = d; best = fid[j]; }
                }
        __syncthreads();
    }
    if (valid) Cl.contact_id[(size_t)f * Cl.n_ids + id] = best;
}

// trace_penetration_after_mpm_kernel :484-510 (after_cloth = 0) / trace_penetration_after_cloth_kernel :520-545 (1).
// One thread per ORIGINAL particle id; inv_cur / inv_prev map ids to slots of frames f / f-1 (nullptr: identity order).
template <class R>
__global__ __launch_bounds__(BLOCK) void k_cloth_trace(DevSim<R> D, int f, int after_cloth, const R* S_cur, const R* S_prev, const int* inv_cur,
                                                       const int* inv_prev, int* warn) {
    const ClothDev& Cl = D.cloth;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= D.N) return;
    const size_t cur = (size_t)f * Cl.n_ids + i, prev = (size_t)(f - 1) * Cl.n_ids + i;
    int pen = after_cloth ? Cl.penetration[cur] : Cl.penetration[prev];
    const int fc = Cl.contact_id[cur], fp = after_cloth ? Cl.contact_before[i] : Cl.contact_id[prev];
    if (fc == -1 || fp == -1) pen = 0;
    else {
        int inverse = 0;
        bool neighbouring = false;
        if (fc != fp) {
            for (int j = 0; j < Cl.n_neighbors; ++j)
                if (Cl.nbr[(size_t)fc * Cl.n_neighbors + j] == fp) {
                    neighbouring = true;
                    inverse = Cl.nbr_dir[(size_t)fc * Cl.n_neighbors + j];
                    break;
                }
        } else neighbouring = true;
        if (neighbouring) {
            double pc[3], pp[3];
            cloth_particle_pos(D, S_cur, inv_cur ? inv_cur[i] : i, pc);
            if (after_cloth) { pp[0] = pc[0]; pp[1] = pc[1]; pp[2] = pc[2]; }
            else cloth_particle_pos(D, S_prev, inv_prev ? inv_prev[i] : i, pp);
            const double* vc = Cl.pos + (size_t)f * Cl.V * 3;
            const double* vp = Cl.pos + (size_t)(f - 1) * Cl.V * 3;
            const int* a = Cl.faces + 3 * fc;
            const int* b = Cl.faces + 3 * fp;
            const bool side_cur = cl_check_side(pc, vc + 3 * a[0], vc + 3 * a[1], vc + 3 * a[2]);
            const bool side_prev = cl_check_side(pp, vp + 3 * b[0], vp + 3 * b[1], vp + 3 * b[2]);
            if ((side_cur == side_prev) == (inverse != 0)) pen = 1 - pen;
        } else atomicAdd(warn, 1);                 // the reference prints "not neighboring faces ... please expand searching region"
    }
    Cl.penetration[cur] = (signed char)pen;
}

// backup_contact_pair_kernel :471-474 and the per-frame copies of copyframe :597-598
__global__ void k_cloth_copy_ids(int N, const int* src_id, int* dst_id, const signed char* src_pen, signed char* dst_pen) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    dst_id[i] = src_id[i];
    if (src_pen) dst_pen[i] = src_pen[i];
}
// check_penetration :555-561
__global__ void k_cloth_count_pen(int N, const signed char* pen, int* total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int c = (i < N && pen[i] == 1) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(total, c);
}

// The particles grid_op_mixed3 :419-428 acts on as the hit list the contact kernels walk: Hit = {slot, 1 | penetration << 1, block, face}.
// contact_id[f, p] >= 0 only says that the particle was inside a face's padded bounding box (1e-2 scale); collide_mixed :236-237 does
// something only within 5e-3 scale of the face (always, for a particle flagged as penetrated: its distance is <= 0).  The list keeps
// the particles that can pass that test (FILTER with a margin; the contact chain repeats the test itself).
template <class R>
__global__ __launch_bounds__(BLOCK) void k_cloth_hit_list(DevSim<R> D, int f) {
    SMAC_CHUNK_PROLOGUE
    if (!valid) return;
    const ClothDev& Cl = D.cloth;
    const size_t at = (size_t)f * Cl.n_ids + D.orig_id[p];
    int face = Cl.contact_id[at];
    const int pen = (face >= 0 && Cl.penetration[at] == 1) ? 1 : 0;
    if (D.collision_type == CONTACT_PARTICLE) D.pmask[p] = face >= 0 ? (1 | (pen << 1)) : 0;   // read back by p2g.grad (rebuilt here when the forward grid comes from a checkpoint)
    // a scene that also holds SDF primitives (round 4, "mixed soft-rigid-cloth"): their band test runs here, so that a particle in reach of both is ONE entry -
    // the contact chain is sequential (primitives in index order, then the sheet).  Hit::mask = sheet bits | primitive band bits << 8.
    int pm = 0;
    if (D.P > 0 && D.collision_type == CONTACT_MIXED) {
        typename pos_of<R>::type xp[3];
        load_pos(frame(D.S, f, D.Npad), D.Npad, p, xp);
        pm = contact_mask(D, f, xp);
    }
    if (face >= 0 && !pen) {
        double px[3];
        cloth_particle_pos(D, frame(D.S, f, D.Npad), p, px);
        const double* vp = Cl.pos + (size_t)f * Cl.V * 3;
        const int* v = Cl.faces + 3 * face;
        if (cl_distance(px, vp + 3 * v[0], vp + 3 * v[1], vp + 3 * v[2]) > (5e-3 + 1e-6) * Cl.par.scale) face = -1;
    }
    if (face < 0 && pm == 0) return;
    Hit h = {p, (face >= 0 ? (1 | (pen << 1)) : 0) | (pm << 8), ch.block, face >= 0 ? face : 0};
    D.hits[hit_slot(D.nhits)] = h;
}

}  // namespace smac
