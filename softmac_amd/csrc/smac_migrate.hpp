// Particle migration between slabs on the device (SURVEY.md 8e "Particle migration"; no reference counterpart).
//
// At a re-sort / env-step boundary a rank hands the particles whose stencil base (mpm_simulator.py:215) left its x-range to the neighbour on
// that side.  Round 2 did this through the host (get_state of the whole slab frame, numpy masks, set_state); here the frame never leaves HBM:
//   k_mig_classify   per slot of frame f: 0 stays, 1 goes left, 2 goes right                      -> three 0/1 flag arrays
//   (hipcub exclusive sums of the flags: stable positions, so a migration is reproducible)
//   k_mig_pack       kept rows -> frame f+1 in its new (identity) order; leaving rows + global ids -> the two send buffers;
//                    records, per NEW index / per sent row, the slot it came from (the backward pass walks this list)
//   (counts and rows travel over RCCL: softmac_hip.hip `migrate`)
//   k_mig_unpack     arriving rows + ids appended behind the kept ones
// and backwards
//   k_mig_grad_keep  adjoint rows of the kept particles back into the slots of frame f
//   k_mig_grad_pack / k_mig_grad_unpack   the adjoint rows of the arrivals go back to where they came from and are added there.
// Rows move as raw scalars of the handle's precision (float32 mode: the fixed-point position words travel bit for bit).
#pragma once
#include "smac_kernels.hpp"

namespace smac {

template <class R>
__global__ void k_mig_classify(int N, int Npad, int n_grid, const R* Sf, int base_lo, int base_hi, int has_left, int has_right, int* keep, int* left, int* right) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N) return;
    typedef typename pos_of<R>::type PX;
    const int b = pos_base(((const PX*)Sf)[poff(q)], n_grid);          // base of make_stencil_pos, x component (row 0)
    const int go = (has_left && b < base_lo) ? 1 : ((has_right && b >= base_hi) ? 2 : 0);
    keep[q] = go == 0; left[q] = go == 1; right[q] = go == 2;
}

// send buffer of one side, `cap` particles: [ids: cap x int64][rows: NCOMP x cap scalars]
template <class R> __device__ __forceinline__ long long* mig_ids(char* buf) { return (long long*)buf; }
template <class R> __device__ __forceinline__ R* mig_rows(char* buf, int cap) { return (R*)(buf + (size_t)cap * sizeof(long long)); }

template <class R>
__global__ void k_mig_pack(int N, int Npad, const R* Sf, R* Sn, const long long* ids_old, long long* ids_new, const int* orig, const int* keep, const int* kpos,
                           const int* lpos, const int* rpos, const int* left, char* buf_l, char* buf_r, int cap_l, int cap_r, int nk, int nl, int* src_slot) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N) return;
    const long long id = ids_old[orig ? orig[q] : q];
    if (keep[q]) {
        const int j = kpos[q];
#pragma unroll 1
        for (int c = 0; c < NCOMP; ++c) Sn[rowoff(c, j, Npad)] = Sf[rowoff(c, q, Npad)];
        ids_new[j] = id;
        src_slot[j] = q;
    } else if (left[q]) {
        const int j = lpos[q];
        R* rows = mig_rows<R>(buf_l, cap_l);
#pragma unroll 1
        for (int c = 0; c < NCOMP; ++c) rows[(size_t)c * cap_l + j] = Sf[rowoff(c, q, Npad)];
        mig_ids<R>(buf_l)[j] = id;
        src_slot[nk + j] = q;
    } else {
        const int j = rpos[q];
        R* rows = mig_rows<R>(buf_r, cap_r);
#pragma unroll 1
        for (int c = 0; c < NCOMP; ++c) rows[(size_t)c * cap_r + j] = Sf[rowoff(c, q, Npad)];
        mig_ids<R>(buf_r)[j] = id;
        src_slot[nk + nl + j] = q;
    }
}

template <class R>
__global__ void k_mig_unpack(int n_in, int cap, int Npad, const char* buf, R* Sn, long long* ids_new, int at) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_in) return;
    const R* rows = mig_rows<R>((char*)buf, cap);
#pragma unroll 1
    for (int c = 0; c < NCOMP; ++c) Sn[rowoff(c, at + j, Npad)] = rows[(size_t)c * cap + j];
    ids_new[at + j] = mig_ids<R>((char*)buf)[j];
}

// padding slots of the new frame (beyond n_new): positions at the middle of the box, everything else zero (as k_rows_from_aos leaves them)
template <class R>
__global__ void k_mig_pad(int n_new, int Npad, R* Sn) {
    const int q = n_new + blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Npad) return;
    typedef typename pos_of<R>::type PX;
    for (int c = 0; c < NCOMP; ++c) Sn[rowoff(c, q, Npad)] = R(0);
    for (int c = 0; c < 3; ++c) ((PX*)Sn)[rowoff(c, q, Npad)] = pos_mid<R>();
}

// backward: A[f][src_slot[j]] += G[j] for the kept particles (G = adjoint of frame f+1 in its identity order)
template <class R>
__global__ void k_mig_grad_keep(int nk, int Npad, const R* G, R* Af, const int* src_slot) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nk) return;
    const int q = src_slot[j];
#pragma unroll 1
    for (int c = 0; c < NCOMP; ++c) Af[rowoff(c, q, Npad)] += G[rowoff(c, j, Npad)];
}
// adjoint rows of the arrivals [at, at + n) -> a send buffer (rows only)
template <class R>
__global__ void k_mig_grad_pack(int n, int cap, int Npad, const R* G, int at, char* buf) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    R* rows = mig_rows<R>(buf, cap);
#pragma unroll 1
    for (int c = 0; c < NCOMP; ++c) rows[(size_t)c * cap + j] = G[rowoff(c, at + j, Npad)];
}
// the adjoint rows of the particles this rank had sent away come back: added into their old slots
template <class R>
__global__ void k_mig_grad_unpack(int n, int cap, int Npad, const char* buf, R* Af, const int* src_slot) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const R* rows = mig_rows<R>((char*)buf, cap);
    const int q = src_slot[j];
#pragma unroll 1
    for (int c = 0; c < NCOMP; ++c) Af[rowoff(c, q, Npad)] += rows[(size_t)c * cap + j];
}

}  // namespace smac
