// Particle binning for the block-sparse grid (no reference counterpart: the reference keeps
// particles in creation order and sweeps a dense grid; see DESIGN.md "layout").
//
//   grid   : cells grouped in 4x4x4 blocks, block-major:  cell = block*64 + (lx*16 + ly*4 + lz)
//            -> one block = 64 consecutive scalars = one 256-B wave access per field
//   ranks  : a particle that is still in the cell of the previous binning KEEPS its rank there (a bit of the cell's 15-bit occupancy word,
//            claimed with one atomicOr); newcomers take the lowest free ranks, then the overflow ranks.  A re-sort therefore moves most
//            particles by a few slots only (bins shift, ranks do not), and the row moves of the frame stay nearly coalesced.
//   sort   : counting sort with key = block*KMAX + min(rank_in_cell, KMAX-1).  All particles of one
//            bin live in DIFFERENT cells, so the 64 lanes of a wave scatter to (mostly) distinct
//            nodes: LDS / global float atomics do not serialise on one address.  Inside a bin the
//            particles are ordered by cell, so neighbouring lanes hit neighbouring tile words
//            (measured: 17 instead of 23.5 LDS cycles per 64-lane f64 atomic, tools/microbench/lds_tile.hip).
//   chunks : each non-empty block is cut into work items of <= 4 waves (256 particles);
//            one workgroup per chunk, one LDS tile per workgroup.
//   active : blocks within [-1,+2]^3 of a block that holds particle bases - every cell a particle can
//            touch until the next re-sort (drift < 4 cells); only these are cleared / swept.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "smac_math.hpp"

namespace smac {

constexpr int KMAX = 16;        // rank-in-cell bins per block
constexpr int CHUNK = 256;      // particles per work item (4 waves)

struct Chunk { int block, start, count; };

__host__ __device__ __forceinline__ int block_of(int nb, int i, int j, int k) { return ((i >> 2) * nb + (j >> 2)) * nb + (k >> 2); }
__host__ __device__ __forceinline__ size_t cell_of(int nb, int i, int j, int k) {
    return (size_t)block_of(nb, i, j, k) * 64 + (((i & 3) << 4) | ((j & 3) << 2) | (k & 3));
}

// per cell, one packed word: bits 0..14 = ranks claimed by particles that stayed in the cell, bits 16.. = number of newcomers
constexpr unsigned RANK_BITS = (1u << (KMAX - 1)) - 1u;
// per particle slot of an epoch: (cell << 5) | min(rank, 31) at the time of its binning
__host__ __device__ __forceinline__ int cellrank_pack(int cell, int r) { return (int)(((unsigned)cell << 5) | (unsigned)(r < 31 ? r : 31)); }

// ---- guests (round 5): a block's particles beyond a whole number of chunks are handed to a face neighbour with room ----
// The fused particle kernels are limited by their workgroup slots: time = chunks x a chunk's lifetime / resident workgroups, and a chunk's lifetime hardly
// depends on how many of its lanes hold a particle.  At 8 particles per cell a 4^3 block holds 400 ... 585: 512 nominal, so three blocks in ten need a THIRD chunk for
// a few dozen particles.  Since the wide tiles (round 4) a particle whose stencil base lies ONE node outside its chunk's block runs on the fast path; so the
// binning may pretend that a particle of the block's outermost cell layer sits in the neighbour's adjacent cell.  Pass 0 counts the blocks' particles, pass 0b lets
// every block whose remainder over whole chunks is small reserve room in a face neighbour that keeps its chunk count, pass 1 (k_sort_rank) hands that many tickets
// to particles of the layer that faces it.  Kernels, lists and maps see an ordinary binning; the guests have three cells of drift left before the halo ends.
constexpr int DONATE_MAX = 96;      // largest remainder a block tries to hand over (its facing layer holds about a quarter of its particles)

template <class R>
__global__ void k_block_count(const R* x0, const R* x1, const R* x2, int N, int n, int nb, int* blk_count) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    typedef typename pos_of<R>::type PX;
    const PX x[3] = {((const PX*)x0)[poff(p)], ((const PX*)x1)[poff(p)], ((const PX*)x2)[poff(p)]};
    int b[3];
    for (int d = 0; d < 3; ++d) {
        const int v = pos_base(x[d], n);
        b[d] = v < 0 ? 0 : (v > n - 3 ? n - 3 : v);
    }
    const int blk = block_of(nb, b[0], b[1], b[2]);
    // the frame is in the order of the last binning: the lanes of a wave mostly share their block - one add for all that share the first lane's
    const unsigned long long act = __ballot(1);
    const int lead = __ffsll((long long)act) - 1;
    const int b0 = __shfl(blk, lead, 64);
    const unsigned long long same = __ballot(blk == b0);
    if (blk != b0) atomicAdd(blk_count + blk, 1);
    else if ((int)(threadIdx.x & 63) == lead) atomicAdd(blk_count + b0, __popcll(same));
}
__device__ __forceinline__ bool donor_candidate(int c) { const int r = c % CHUNK; return c > CHUNK && r > 0 && r <= DONATE_MAX && r * 6 <= c; }
// direction d = 2 * axis + (0: towards +, 1: towards -)
__global__ void k_donate_plan(int nblocks, int nb, const int* blk_count, int* incoming, int* don_dir, int* don_left) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    don_dir[b] = -1;
    don_left[b] = 0;
    const int c = blk_count[b];
    if (!donor_candidate(c)) return;
    const int r = c % CHUNK;
    const int q[3] = {b / (nb * nb), (b / nb) % nb, b % nb};
    for (int t = 0; t < 6; ++t) {
        const int d = (t + b) % 6, ax = d >> 1, sg = (d & 1) ? -1 : 1;
        int m[3] = {q[0], q[1], q[2]};
        m[ax] += sg;
        if (m[ax] < 0 || m[ax] >= nb) continue;
        const int nbr = (m[0] * nb + m[1]) * nb + m[2];
        const int cn = blk_count[nbr];
        if (cn <= 0 || donor_candidate(cn)) continue;                                // (a block that gives does not take: its own remainder would only be replaced)
        const int room = (cn + CHUNK - 1) / CHUNK * CHUNK - cn;
        if (room < r) continue;
        if (atomicAdd(incoming + nbr, r) + r <= room) { don_dir[b] = d; don_left[b] = r; return; }
        atomicSub(incoming + nbr, r);
    }
}

// pass 1: claim a rank in the particle's cell (see "ranks" above); the cell and the claim go to pass 2.
// Also records the largest velocity component (float bits of a non-negative value order like unsigned ints): the
// host turns it into the number of substeps the binning stays valid for.
template <class R>
__global__ void k_sort_rank(const R* x0, const R* x1, const R* x2, const R* v0, const R* v1, const R* v2, int N, int n, int nb,
                            R inv_dx, int* cell_count, int* cell_out, int* tag_out, float* vmax_part, const int* cellrank_old,
                            const int* don_dir = nullptr, int* don_left = nullptr) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    float vm = 0.f;
    if (p < N) {
        const size_t po = poff(p);
        const float a = fabsf((float)v0[po]), b = fabsf((float)v1[po]), c = fabsf((float)v2[po]);
        vm = fmaxf(a, fmaxf(b, c));
    }
    for (int o = 32; o > 0; o >>= 1) vm = fmaxf(vm, __shfl_xor(vm, o, 64));
    // per-workgroup maxima go to an array (k_bin_masks folds it): thousands of atomics on one address would cost ~35 ns each
    __shared__ float wg_max[16];
    if ((threadIdx.x & 63) == 0) wg_max[threadIdx.x >> 6] = vm;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = wg_max[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, wg_max[w]);
        vmax_part[blockIdx.x] = m;
    }
    if (p >= N) return;
    typedef typename pos_of<R>::type PX;                 // position rows: smac_math.hpp pos_of
    const PX x[3] = {((const PX*)x0)[poff(p)], ((const PX*)x1)[poff(p)], ((const PX*)x2)[poff(p)]};
    (void)inv_dx;
    int b[3];
    for (int d = 0; d < 3; ++d) {
        int v = pos_base(x[d], n);                       // the base of make_stencil_pos
        b[d] = v < 0 ? 0 : (v > n - 3 ? n - 3 : v);
    }
    if (don_dir) {                                                                   // guests: see k_donate_plan
        const int blk = block_of(nb, b[0], b[1], b[2]);
        const int d = don_dir[blk];
        if (d >= 0) {
            const int ax = d >> 1, sg = (d & 1) ? -1 : 1;
            const int to = b[ax] + sg;
            if ((b[ax] & 3) == (sg > 0 ? 3 : 0) && to >= 0 && to <= n - 3 && don_left[blk] > 0 && atomicSub(don_left + blk, 1) > 0) b[ax] = to;
        }
    }
    const int cell = (int)cell_of(nb, b[0], b[1], b[2]);
    int tag;
    const int cr = cellrank_old ? cellrank_old[p] : -1;
    if (cr >= 0 && (int)((unsigned)cr >> 5) == cell && (cr & 31) < KMAX - 1) {
        tag = cr & 31;                                   // stayed: keeps its rank (two stayers of a cell never held the same one)
        atomicOr((unsigned*)cell_count + cell, 1u << tag);
    } else {
        tag = 16 + (int)(atomicAdd((unsigned*)cell_count + cell, 1u << 16) >> 16);   // newcomer number w (also: ranks beyond the bins, first binning)
    }
    cell_out[p] = cell;
    tag_out[p] = tag;
}

// ranks of a cell from its packed word: the newcomers take the lowest free ranks below KMAX-1, the rest overflow
__device__ __forceinline__ void cell_ranks(unsigned cc, unsigned& occ, int& over) {
    const unsigned stay = cc & RANK_BITS;
    const int narr = (int)(cc >> 16);
    unsigned fb = ~stay & RANK_BITS;
    const int nfree = __popc(fb);
    const int fill = narr < nfree ? narr : nfree;
    unsigned taken = 0;
    for (int i = 0; i < fill; ++i) { const unsigned low = fb & (0u - fb); taken |= low; fb ^= low; }
    occ = stay | taken;
    over = narr - fill;
}

// pass 1b, one wave per block: bin sizes and, per bin, the set of cells that own a particle of that rank
// (bit c of mask[block*KMAX + r] <=> cell c holds a particle of rank r).  A particle's slot inside its bin is
// the number of lower cells in the mask; the overflow bin (rank >= KMAX-1) uses a prefix sum of the excess counts.
__global__ void k_bin_masks(int nblocks, const int* cell_count, int* bin_count, unsigned long long* mask, int* over_prefix,
                            const float* vmax_part, int nparts, float* vmax_out) {
    if (blockIdx.x == gridDim.x - 1) {                           // fold the per-workgroup speed maxima of k_sort_rank
        __shared__ float fold[16];
        float m = 0.f;
        for (int i = threadIdx.x; i < nparts; i += blockDim.x) m = fmaxf(m, vmax_part[i]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) fold[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, fold[w]);
            *vmax_out = m;
        }
    }
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= nblocks) return;
    const int lane = threadIdx.x & 63;
    const unsigned cc = (unsigned)cell_count[(size_t)b * 64 + lane];
    if (__ballot(cc != 0u) == 0ull) {
        if (lane < KMAX) bin_count[b * KMAX + lane] = 0;
        return;
    }
    unsigned occ;
    int over;
    cell_ranks(cc, occ, over);
    for (int r = 0; r < KMAX - 1; ++r) {
        const unsigned long long m = __ballot((occ >> r) & 1u);
        if (lane == 0) { mask[b * KMAX + r] = m; bin_count[b * KMAX + r] = __popcll(m); }
    }
    int incl = over;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    over_prefix[(size_t)b * 64 + lane] = incl - over;
    if (lane == 63) bin_count[b * KMAX + KMAX - 1] = incl;
}

// pass 2 (after an exclusive scan of bin_count): final rank, destination index, composed original id, (cell, rank) of the new slot
__global__ void k_sort_dest(int N, const int* cell_in, const int* tag_in, const int* cell_count, const int* bin_start, const unsigned long long* mask,
                            const int* over_prefix, const int* orig_old, int* dest, int* orig_new, int* cellrank_new) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const int cell = cell_in[p], tag = tag_in[p], cl = cell & 63, blk = cell >> 6;
    int r;
    if (tag < KMAX - 1) r = tag;
    else {
        const int w = tag - 16;
        unsigned fb = ~((unsigned)cell_count[cell]) & RANK_BITS;
        const int nfree = __popc(fb);
        if (w < nfree) {
            for (int i = 0; i < w; ++i) fb &= fb - 1u;
            r = __ffs((int)fb) - 1;
        } else r = (KMAX - 1) + (w - nfree);
    }
    const int k = blk * KMAX + (r < KMAX - 1 ? r : KMAX - 1);
    int pos;
    if (r < KMAX - 1) pos = __popcll(mask[k] & ((1ull << cl) - 1ull));
    else pos = over_prefix[(size_t)blk * 64 + cl] + (r - (KMAX - 1));
    const int q = bin_start[k] + pos;
    dest[p] = q;
    orig_new[q] = orig_old ? orig_old[p] : p;
    if (cellrank_new) cellrank_new[q] = cellrank_pack(cell, r);
}

// particles the last binning moved by more than `farther_than` slots (smac_get_param "resort_moved" / "resort_far": on request only, one atomic per workgroup)
__global__ void k_count_moved(int N, const int* dest, unsigned long long* out, int farther_than) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int dd = p < N ? dest[p] - p : 0;
    const int moved = (dd > farther_than || dd < -farther_than) ? 1 : 0;
    const int n = __syncthreads_count(moved);
    if (threadIdx.x == 0 && n) atomicAdd(out, (unsigned long long)n);
}

// Host IO: f64 AOS arrays in the caller's particle order <-> component rows of a frame in the order of an epoch
// (orig[q] = caller's id of the particle in slot q; nullptr = identity).  `ident` 1: the rows hold F - I, the host sees F;
// `ident` 2: position rows (pos_of<R>: fixed point in f32 mode), padding slots sit at the middle of the box.
template <class R>
__global__ void k_rows_from_aos(int N, int Npad, const double* src, int stride, int offset, int cnt, const int* orig, int ident, R* rows) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Npad) return;
    const size_t id = q < N ? (size_t)(orig ? orig[q] : q) : 0;
    if (ident == 2) {
        typedef typename pos_of<R>::type PX;
        for (int c = 0; c < cnt; ++c) {
            PX v;
            pos_set(q < N ? src[id * stride + offset + c] : 0.5, v);
            ((PX*)rows)[rowoff(c, q, Npad)] = v;
        }
        return;
    }
    for (int c = 0; c < cnt; ++c) {
        const double sub = (ident && (c == 0 || c == 4 || c == 8)) ? 1.0 : 0.0;
        rows[rowoff(c, q, Npad)] = q < N ? (R)(src[id * stride + offset + c] - sub) : R(0);
    }
}
template <class R>
__global__ void k_rows_add_aos(int N, int Npad, const double* src, int cnt, const int* orig, R* rows) {     // rows[c][q] += src[id][c]
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N) return;
    const size_t id = (size_t)(orig ? orig[q] : q);
    for (int c = 0; c < cnt; ++c) rows[rowoff(c, q, Npad)] += (R)src[id * cnt + c];
}
template <class R>
__global__ void k_rows_to_aos(int N, int Npad, const R* rows, int cnt, const int* orig, int ident, double* dst, int stride, int offset) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N) return;
    const size_t id = (size_t)(orig ? orig[q] : q);
    if (ident == 2) {
        typedef typename pos_of<R>::type PX;
        for (int c = 0; c < cnt; ++c) dst[id * stride + offset + c] = pos_get(((const PX*)rows)[rowoff(c, q, Npad)]);
        return;
    }
    for (int c = 0; c < cnt; ++c) {
        const double add = (ident && (c == 0 || c == 4 || c == 8)) ? 1.0 : 0.0;
        dst[id * stride + offset + c] = (double)rows[rowoff(c, q, Npad)] + add;
    }
}

// pass 3: move one component row of a frame
template <class R>
__global__ void k_sort_move(int N, const int* dest, const R* src, R* dst, int Npad, int ncomp) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const int q = dest[p];
    const size_t pq = poff(q), pp = poff(p);
    for (int c = 0; c < ncomp; ++c) dst[rowbase(c, Npad) + pq] = src[rowbase(c, Npad) + pp];
}

// gather form, used to bring an adjoint frame from one epoch's order into another's:
//   dst[q] = src[ map[q] ]
template <class R>
__global__ void k_gather_rows(int N, const int* map, const R* src, R* dst, int Npad, int ncomp) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N) return;
    const int p = map[q];
    const size_t pq = poff(q), pp = poff(p);
    for (int c = 0; c < ncomp; ++c) dst[rowbase(c, Npad) + pq] = src[rowbase(c, Npad) + pp];
}

// inverse[orig[q]] = q
__global__ void k_invert(int N, const int* orig, int* inv) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < N) inv[orig[q]] = q;
}
// map[q_to] = inv_from[ orig_to[q_to] ]  : position in epoch "from" of the particle at q_to in epoch "to"
__global__ void k_compose(int N, const int* orig_to, const int* inv_from, int* map) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < N) map[q] = inv_from[orig_to[q]];
}

// what the host needs of a re-sort, in ONE write to pinned host memory: the chunk and active-block totals, the fastest particle's speed, the drift
// flags of the epoch that ends (four hipMemcpyAsync to pageable memory cost 25 us of host round trip each, the GPU idle: profiles/r04_ah_sort_host.txt)
__global__ void k_sort_info(const int* chunk_total, const int* active_total, const unsigned* vmax, const int* drift, int read_drift, int* host_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    host_out[0] = *chunk_total; host_out[1] = *active_total; host_out[2] = (int)*vmax;
    for (int i = 0; i < 4; ++i) host_out[3 + i] = read_drift ? drift[i] : 0;      // (the host reads after synchronising with the stream)
}

// per block: particle count (from the scanned bins), number of chunks, halo flags
__global__ void k_block_info(int nblocks, int nb, const int* bin_start, int N, int* block_start, int* block_chunks,
                             int* active_flag) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const int s = bin_start[b * KMAX];
    const int e = (b + 1 < nblocks) ? bin_start[(b + 1) * KMAX] : N;
    const int cnt = e - s;
    block_start[b] = s;
    const int waves = (cnt + 63) / 64;
    block_chunks[b] = (waves + 3) / 4;
    if (cnt > 0) {
        const int bz = b % nb, by = (b / nb) % nb, bx = b / (nb * nb);
        for (int i = -1; i <= 2; ++i)
            for (int j = -1; j <= 2; ++j)
                for (int k = -1; k <= 2; ++k) {
                    const int X = bx + i, Y = by + j, Z = bz + k;
                    if (X < 0 || Y < 0 || Z < 0 || X >= nb || Y >= nb || Z >= nb) continue;
                    active_flag[(X * nb + Y) * nb + Z] = 1;
                }
    }
}

// write the chunk list (chunk_start = exclusive scan of block_chunks) and the compacted active list
// (also: the epoch's own copies of the four per-block tables - four device-to-device copies of 128 KB each cost the host 8 us apiece right after the
//  re-sort's synchronisation, with the GPU idle; `chunk_cap`: the chunk list's capacity - the host checks the total AFTER this kernel has run)
__global__ void k_emit_lists(int nblocks, int N, const int* bin_start, const int* block_start, const int* block_chunks,
                             const int* chunk_start, const int* active_flag, const int* active_start, Chunk* chunks,
                             int* active, int chunk_cap, int* ep_chunk_start, int* ep_block_chunks, int* ep_block_active, int* ep_block_slot) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= nblocks) {
        ep_chunk_start[b] = chunk_start[b]; ep_block_chunks[b] = block_chunks[b];
        ep_block_active[b] = active_flag[b]; ep_block_slot[b] = active_start[b];
    }
    if (b >= nblocks) return;
    const int nch = chunk_start[b] + block_chunks[b] <= chunk_cap ? block_chunks[b] : 0;
    if (nch > 0) {
        const int s = block_start[b];
        const int e = (b + 1 < nblocks) ? bin_start[(b + 1) * KMAX] : N;
        const int waves = (e - s + 63) / 64;
        int w0 = 0;
        for (int c = 0; c < nch; ++c) {
            const int w = waves / nch + (c < waves % nch ? 1 : 0);       // waves of this chunk (<= 4)
            Chunk ch;
            ch.block = b;
            ch.start = s + w0 * 64;
            const int end = s + (w0 + w) * 64;
            ch.count = (end < e ? end : e) - ch.start;
            chunks[chunk_start[b] + c] = ch;
            w0 += w;
        }
    }
    if (active_flag[b]) active[active_start[b]] = b;
}

// Tail reduction (round 5, smac_kernels.hpp tail_arrive): a block B receives scatter from the chunks of the blocks B + [-1, 1]^3 (a chunk's wide LDS tile
// reaches one node past its block on either side); expect[B] = that many chunks.  The chunk whose arrival completes the count sums B's slab records.
__global__ void k_tail_expect(int nblocks, int nb, const int* block_chunks, int* expect) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const int bz = b % nb, by = (b / nb) % nb, bx = b / (nb * nb);
    int n = 0;
    for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j)
            for (int k = -1; k <= 1; ++k) {
                const int X = bx + i, Y = by + j, Z = bz + k;
                if (X < 0 || Y < 0 || Z < 0 || X >= nb || Y >= nb || Z >= nb) continue;
                n += block_chunks[(X * nb + Y) * nb + Z];
            }
    expect[b] = n;
}

}  // namespace smac
