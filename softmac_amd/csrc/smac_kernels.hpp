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
#include <type_traits>
#include "smac_math.hpp"
#include "smac_cloth.hpp"
#include "smac_sort.hpp"

namespace smac {

// A/B on MI355X (tools/ab.sh): k_g2p_grad is fastest at 2 waves/SIMD without spills; k_p2g_grad at 3 with the small
// LDS stash (3 workgroups per CU); forcing more occupancy only buys scratch traffic.
#ifndef SMAC_OCC_HEAVY
#define SMAC_OCC_HEAVY 2
#endif
#ifndef SMAC_OCC_G2PG
#define SMAC_OCC_G2PG 3
#endif
#ifndef SMAC_OCC_P2GG
#define SMAC_OCC_P2GG 3
#endif
#ifndef SMAC_OCC_G2P
#define SMAC_OCC_G2P 4
#endif
#ifndef SMAC_OCC_P2G
#define SMAC_OCC_P2G 6
#endif
#ifndef SMAC_G2P_ROLLED
#define SMAC_G2P_ROLLED 1        // the G2P gather with its x-planes as a real loop: 9 records live instead of 27, k_g2p 115 -> 88 and k_g2p_p2g 121 -> 96 VGPRs, 5 waves per SIMD (profiles/r04_ad_forward_variants.txt; 0: the unrolled form)
#endif
#ifndef SMAC_CONTACT_HYBRID
#define SMAC_CONTACT_HYBRID 1    // float32 mode: forecast contact in two widths - the signed distance in f64, the rest in f32 (collide_mixed_hybrid; 0: all f64, rounds 1-4)
#endif
#ifndef SMAC_HITS_PER_WAVE
#define SMAC_HITS_PER_WAVE 2     // contact kernels: hits per 64-lane wave (two 32-lane groups).  1 was tried in round 5 on the theory that the two hits of a wave are often in reach of
#endif                           // DIFFERENT primitives and the wave then runs both collide_mixed chains: k_contact_hits 19.1 -> 21.0 us, k_contact_grad 25.7 -> 33.2 us (profiles/r05_contact.txt) -
                                 // twice the workgroups pay twice the per-workgroup tile zeroing / flush; the chains are not what these kernels wait for
#ifndef SMAC_TAIL_BUILD
#define SMAC_TAIL_BUILD 0        // 1: the particle kernels carry the tail reduction (tail_arrive; run-time switch SMAC_TAIL_REDUCE).  Built, parity-green and 15 % SLOWER than the
#endif                           // reduction launches it replaces (profiles/r05_tail_reduce.txt): the shipped kernels are compiled without it
#ifndef SMAC_PHASE_CLOCK
#define SMAC_PHASE_CLOCK 0       // 1 (tools/phase_clock.sh only): 1 workgroup in 16 of the particle kernels files s_memtime at its phase boundaries
#endif
// arithmetic type of the constitutive model (SVD, stress, their adjoint) inside the particle kernels: the storage type R,
// or double whatever R is (SMAC_CONST_F64=1; measured for the f32 accuracy study, tools/prec_probe.py)
#ifndef SMAC_CONST_F64
#define SMAC_CONST_F64 0
#endif
template <class R> struct const_t { typedef typename std::conditional<SMAC_CONST_F64 != 0, double, R>::type type; };
constexpr int BLOCK = 256;
enum { CX = 0, CV = 3, CC = 6, CF = 15, NCOMP = 24 };
static_assert(NCOMP == NCOMP_ROWS, "frame layout helpers (smac_math.hpp) assume 24 components");

template <class R> struct alignas(4 * sizeof(R)) Vec4 { R x, y, z, w; };
struct Hit { int p, mask, block, pad; };

// the shared x-planes of the slab decomposition: entry s of the list = buffer slot (0 = left neighbour, 1 = right neighbour) and first plane
struct HaloSides { int count; int slot[2]; int plane0[2]; };

template <class R> struct DevSim {
    int N, Npad, n, P, n_control, substeps, collision_type, sticky, max_frames;
    R dt, inv_dx, dx, p_mass, stress_scale;
    R m_eps;                     // a grid node carries a velocity when its mass exceeds this (1e-10, mpm_simulator.py:286; rescaled with mpm_scale^-2)
    R g[3];
    Material<R> mat;
    // Two-entry material table (round 4, BASELINE config C5 "two material blocks"): the reference's mu / lam / yield_stress are PER-PARTICLE fields
    // (mpm_simulator.py:47-49, filled uniformly at :86-90); a scene with two kinds of particles keeps two (mu, lam, yield) entries and one selector byte
    // per particle (indexed by ORIGINAL particle id, like control_idx) instead of three rows.  mat_id == nullptr: one material (the MAT2 = false kernels).
    R mu2, lam2, yield_c2;
    const unsigned char* mat_id;
    R* S;
    R* A;
    R* Af;                       // adjoint frame of the substep being reversed (frames may live in rolling slots: set per launch)
    R* Af_prev;                  // adjoint frame f - 1 (k_p2g_g2p_grad only)
    // grid: one 4-scalar record per cell and field, so a stencil node is ONE 16-byte access
    Vec4<R> *vin, *vmix, *vout;         // {m, p_x, p_y, p_z} / {v_mixed, 0} / {v_out, 0}
    Vec4<R> *ain, *amix, *aout;         // adjoints: {grid_m.grad, grid_v_in.grad} / {grid_v_mixed.grad, 0} / {grid_v_out.grad, 0}
    // {m, p} that P2G adds with global atomics - the shell of a chunk's wide tile, the stencil of a particle more than a cell past its block - goes to a field
    // of its own: k_grid_op adds it to the slabs' sum, writes {m, p} to vin and leaves vdrift zero again.  So P2G of substep f + 1 may run while the
    // checkpoint save of substep f still reads vin (k_g2p_p2g), and vin needs no clear pass.  Invariant: all zero outside [P2G, grid_op].
    Vec4<R>* vdrift;
    PrimTable<R> prim[MAX_PRIMS];        // tables in R: band filter, collision types 0 / 1
    PrimTable<double> prim64[MAX_PRIMS]; // tables in f64: the forecast contact chain (k_contact_hits / k_contact_grad) runs in double
    // rigid-body state, its adjoint and the wrench accumulators are f64 whatever R is: a pose rounded to float (3e-8 at 0.5)
    // would be divided by dt in the forecast push-out like a position error
    double* prim_state;
    double* prim_grad;
    double* ext_f;
    double* ext_f_grad;
    double dt64;
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
    const R* An;                 // adjoint of frame f+1 in THIS epoch's particle order (A[f+1] or a re-ordered copy) ...
    const int* An_map;           // ... or (round 4) A[f+1] as it lies, binned differently: particle p's rows are at An_map[p] (nullptr: at p).  The backward sweep
                                 // crosses a re-sort through the sort's own destination map INSIDE the two kernels that read frame f+1's adjoint, instead of
                                 // a gather pass over the whole frame first (k_gather_rows: 24 rows read + 24 written, 40-76 us per crossing)
    Vec4<R>* slab;               // [nchunks][TILE_WORDS] per-chunk tiles (P2G: {m,p} ; G2P adjoint: {grid_v_out.grad,0})
    struct Hit* hits;            // particles inside a contact band this frame (written by k_p2g)
    int* nhits;
    int* cand;                   // (unused since the contact kernels walk the hit list; kept for the counter pair)
    int* ncand;
    int* last_counts;            // {nhits, ncand} of the last forward substep (k_g2p empties the lists)
    int* pmask;                  // per particle slot: bit i = inside primitive i's band (collision_type 1 only)
    int any_contact;
    int cur_frame;               // frame of the substep being processed (kernels without an f argument)
    const int* block_chunk_start;   // per block: first chunk / number of chunks (dense, nb^3)
    const int* block_chunks;
    int* drift_flag;
    int frame_shift;             // duplicate frames before this segment (smac_set_segment): physical substep = f - frame_shift
    int check_next;              // k_g2p: frame f+1 is processed with THIS binning too, so x[f+1] must still lie inside the halo
    int open_x;                  // slab decomposition: bit 0 / bit 1 = no wall at the low / high x end (neighbour slab there)
    int slab_base_lo, slab_base_hi;   // slab decomposition: stencil bases (x) this rank's grid planes can take deposits for, drift tolerance included
                                      // (a particle beyond them would scatter onto planes nobody exchanges: flagged by k_g2p, drift_flag[2]); lo > hi: no check
    const int* block_active;     // dense per-block flag of the current epoch (halo packing)
    const int* block_slot;       // dense per block: its slot in the active list of the current epoch (valid where block_active)
    // k_g2p<R, true>: the frame's checkpoint save ([active slot][vin | vmix | vout][64 cells]) rides in the same launch (its first
    // `save_blocks` workgroups, a multiple of 8)
    Vec4<R>* save_ck;
    struct Hit* save_hits;
    int* save_nhits;
    int* save_nhits_host;
    int save_hit_cap;
    int save_blocks;
    // One byte per active slot of the frame whose checkpoint this launch files or restores (nullptr: off): 1 = the block held no mass - {m,p} all zero,
    // hence v_mixed = v_out = 0 - and nothing was filed for it.  The drift margin around the cloud is a third of the active blocks (round 3: the grid
    // passes run at the bandwidth of their 16-byte accesses, so bytes not moved are time).  ck_flags_next: the frame k_reduce_grid_grad_ahead restores.
    unsigned char* ck_flags;
    const unsigned char* ck_flags_next;
    int fk_ride;                 // > 0: this launch also carries forward_kinematics (k_g2p: its last workgroup) / its adjoint (the grid-adjoint reduction: its last
    size_t fk_stride;            // fk_ride workgroups) of that many velocity-controlled primitives (primitive_base.py:280-283, mpm_simulator.py:329-331, 367-369)
    int* nhits_next;             // the hit counter of the NEXT substep (two counters alternate by frame parity): emptied here while this frame's is still read
    struct Hit* hits_next;       // ... and the NEXT substep's hit list (k_g2p_p2g: P2G of substep f + 1 appends to it while the save part copies this substep's)
    // The library's slab loop, forward {m,p} exchange (round 4): k_grid_op's first piece writes the shared planes' records straight into the send buffer
    // (k_halo_pack2's work: dense (np, n, n) per side, cells of inactive blocks stay zero from the buffer's reset at the epoch change), its second piece
    // adds the received planes (k_halo_unpack_add2's work) - two launches less per substep.  halo_hs.count = 0: off.
    HaloSides halo_hs;
    int halo_np;
    Vec4<R>* halo_send;
    const Vec4<R>* halo_recv;
    int keep_vmix;               // k_grid_op also stores grid_v_mixed (the slab phases' halo exchange sends v_out - v_mixed); otherwise it is recomputed where it is read
    int zero_next_hits;          // k_grid_op: empty the next substep's counter (its P2G rides in this substep's G2P launch and appends right away)
    // Tail reduction (round 5): the slab reduction + grid_op (forward) / + grid_op's node adjoint (backward) are done INSIDE the particle
    // kernel that scatters, by the workgroup whose arrival completes a block's count - k_grid_op and the reduction half of k_reduce_grid_grad_ahead go.
    int tail_on;                 // this launch's scatter ends with the arrival protocol (tail_arrive)
    int tail_rule;               // handle-wide: binning valid only while every stencil stays inside the 27-block neighbourhood of the particle's chunk (g2p_particle)
    int tail_extra;              // arrivals per block on top of tail_expect (1: the checkpoint-save wave of the block in the same launch reads what the reducer overwrites)
    int* tail_cnt;               // [blocks] arrival counters, zero between launches (the last arriver resets its block's)
    const int* tail_expect;      // [blocks] chunks of the 27 blocks around it (smac_sort.hpp k_tail_expect)
    ClothDev cloth;              // soft <-> cloth contact (present = 0: none)
};

// LDS tile: the 6x6x6 nodes a particle whose base lies in a 4x4x4 block can touch (origin = 4*block).
// A chunk accumulates its scatter there with LDS atomics and stores the tile ONCE, coalesced, to its
// slab in HBM; the grid kernels then sum, per node, the <= 8 slabs that overlap it.  Particles that
// drifted out of their block since the last sort fall back to global atomics on the dense arrays.
constexpr int TW = 6, TSY = 6, TSX = 36, TILE_WORDS = 216;
constexpr int CK_WORDS = 128;    // grid checkpoint: [active slot][{m,p} | v_out][64 cells] (v_mixed is recomputed from {m,p}: grid_v_mixed_at)
__device__ __forceinline__ int tile_index(int li, int lj, int lk) { return li * TSX + lj * TSY + lk; }
// Order of a chunk's 216 records in its SLAB (round 5): by DESTINATION block.  The 6^3 core of a tile falls on 8 grid blocks - region q = 4 ex + 2 ey + ez holds the
// nodes whose local coordinate is >= 4 along the axes with e = 1 (they belong to the +1 block there) - and a region's nodes are stored in the destination block's
// own cell order.  The wave that sums a block reads, from each overlapping chunk, ONE contiguous run of 64 / 32 / 16 / 8 records (1 KB ... 128 B, every run a
// multiple of 128 B from the slab's start); in the tile's own row-major order the same wave read 4-record (64-B) runs: half of every 128-B line fetched for nothing
// (PMC: 31.6 MB read for 18 MB of records in k_grid_op).  MEASURED (profiles/r05_slab_order.txt): k_grid_op 12.9 -> 12.3 us, the backward reduction 14.9 -> 13.9 us - and the fused
// backward particle kernel, which stores its tile in that order, 121.9 -> 123.5 us: nothing in all (3,575 vs 3,568 substeps/s).  Default 0: the tile's own order (rounds 1-4).
#ifndef SMAC_SLAB_BY_BLOCK
#define SMAC_SLAB_BY_BLOCK 0
#endif
__device__ __forceinline__ int slab_region_offset(int q) {          // 64, 32, 32, 16, 32, 16, 16, 8 records
    return q == 0 ? 0 : (q == 1 ? 64 : (q == 2 ? 96 : (q == 3 ? 128 : (q == 4 ? 144 : (q == 5 ? 176 : (q == 6 ? 192 : 208))))));
}
// record of destination-local cell (lx, ly, lz) in region q (requires lx < 2 where ex, etc.)
__device__ __forceinline__ int slab_record(int q, int lx, int ly, int lz) {
#if SMAC_SLAB_BY_BLOCK
    const int sy = (q & 2) ? 1 : 2, sz = (q & 1) ? 1 : 2;
    return slab_region_offset(q) + (((lx << sy) + ly) << sz) + lz;
#else
    return tile_index(lx + 4 * (q >> 2), ly + 4 * ((q >> 1) & 1), lz + 4 * (q & 1));
#endif
}
// tile node (0..5 per axis) of slab record i
__device__ __forceinline__ void slab_node(int i, int& ti, int& tj, int& tk) {
#if SMAC_SLAB_BY_BLOCK
    const int q = i < 64 ? 0 : (i < 96 ? 1 : (i < 128 ? 2 : (i < 144 ? 3 : (i < 176 ? 4 : (i < 192 ? 5 : (i < 208 ? 6 : 7))))));
    const int r = i - slab_region_offset(q);
    const int sy = (q & 2) ? 1 : 2, sz = (q & 1) ? 1 : 2;                 // log2 of the region's extent along y, z (4 or 2 nodes): shifts, not divisions
    ti = (r >> (sy + sz)) + 4 * (q >> 2); tj = ((r >> sz) & ((1 << sy) - 1)) + 4 * ((q >> 1) & 1); tk = (r & ((1 << sz) - 1)) + 4 * (q & 1);
#else
    ti = i / TSX; tj = (i / TSY) % TW; tk = i % TW;
#endif
}
// WIDE tile of the particle kernels (round 4): the same 6^3 plus ONE node of slack on either side, 8x8x8 nodes with origin 4*block - 1.  A particle that
// crossed a block face since the last sort (they jitter across the faces: 1-2 % of them by the end of a re-sort interval, i.e. a lane in nearly EVERY
// wave) still gathers from LDS and scatters into LDS; the chunk's 6^3 core goes to its slab as before and the non-zero sums on the shell go to the dense
// field with one global atomic per word (tile_flush_shell), where the grid kernels pick them up as the "drift fallback part".  Only a lane that moved
// more than a cell past its block sends its wave through the per-node LDS / global form.  Rows are padded to 10 records: with 8 the four y-rows a group
// of 16 lanes reads with one ds_read_b128 would fall on the same 16 banks (10: 2 lanes per bank, as with the 6-wide rows before).
#ifndef SMAC_WIDE_TILE
#define SMAC_WIDE_TILE 1
#endif
constexpr int PW = SMAC_WIDE_TILE ? 8 : 6, PO = SMAC_WIDE_TILE ? 1 : 0, PSY = SMAC_WIDE_TILE ? 10 : 6, PSX = SMAC_WIDE_TILE ? 80 : 36,
              PTILE = SMAC_WIDE_TILE ? 640 : 216;

// LDS tiles accumulate in f64 whatever R is: measured on gfx950 (tools/microbench/lds_atomics.hip),
// ds_add_f64 retires a conflict-free wave instruction in ~10 cycles while ds_add_f32 is lane-serial
// (~193 cycles) - and the f64 sum is the more accurate one anyway.
typedef double tile_t;

// Scatter tiles of the two hot scatter kernels (k_p2g, k_g2p_grad).
//   R = double : f64 words, ds_add_f64.
//   R = float  : 32-bit FIXED-POINT words, ds_add_u32 - 5.5 instead of 17 LDS cycles per 64-lane atomic
//                (tools/microbench/lds_tile.hip; ds_add_f32 is lane-serial on gfx950 and not an option).  The scale is
//                chosen per chunk from a bound B on any single contribution (quadratic B-spline: w <= 0.75^3 and
//                w |offset - fx| <= 0.25 * 0.75^2 per axis): a contribution is rounded to B * 2^-23 - f32-grade
//                accumulation, as the reference's own f32 atomics would be -, a node sum of the <= 256 contributions
//                of a chunk cannot overflow, and integer adds make the tile independent of the order of arrival.
template <class R> struct ScatterTile { typedef double word; };
template <> struct ScatterTile<float> { typedef int word; };
constexpr float FIX_RANGE = 8388000.0f;      // < 2^31 / 256: |q| <= FIX_RANGE per contribution, 256 of them per node at most
constexpr float W_MAX = 0.421875f;           // 0.75^3
constexpr float WD_MAX = 0.140625f;          // 0.25 * 0.75^2
__device__ __forceinline__ void tile_add(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void tile_add(double* p, float v) { __hip_atomic_fetch_add(p, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// Fixed point resolves a node sum to 2^-23 of the CHUNK's largest contribution.  That is f32-grade where a node collects
// many contributions, but a node that only sees weights of 1e-4 (isolated particles, spray, the rim of a thin sheet) keeps
// 1e-3 relative precision, and grid_op divides by such a mass (its adjoint by its square).  Chunks of at most SPARSE_MAX
// particles - where that regime lives, and where LDS atomics are not the bottleneck - accumulate in f64 words instead.
constexpr int SPARSE_MAX = 128;
__device__ __forceinline__ void tile_add(int* p, float v) {
    __hip_atomic_fetch_add(p, __float2int_rn(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// largest `bound` over the workgroup (every thread calls it; 0 for idle lanes) -> scale to / from tile units
template <class R> __device__ __forceinline__ void tile_scale(R bound, R* scratch4, R& to_tile, R& from_tile);
template <> __device__ __forceinline__ void tile_scale<double>(double, double*, double& to_tile, double& from_tile) {
    to_tile = from_tile = 1.0;
    __syncthreads();
}
// max of a non-negative float over the wave, on the DPP crossbar: the bit pattern of a non-negative float orders like an integer, a butterfly of
// two quad permutes and the two row mirrors leaves each row's maximum in all of its 16 lanes, four v_readlane + s_max join the rows.  (__shfl_xor
// compiled to six ds_bpermute_b32, each followed by s_waitcnt lgkmcnt(0): six LDS round trips in front of the barrier of every workgroup.)
__device__ __forceinline__ float wave_max_nonneg(float v) {
    int x = __float_as_int(v);
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false));   // row_half_mirror
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false));   // row_mirror
    const int a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16), c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
    return __int_as_float(max(max(a, b), max(c, d)));
}
template <> __device__ __forceinline__ void tile_scale<float>(float bound, float* scratch4, float& to_tile, float& from_tile) {
    bound = wave_max_nonneg(bound);
    if ((threadIdx.x & 63) == 0) scratch4[threadIdx.x >> 6] = bound;
    __syncthreads();
    const float B = fmaxf(fmaxf(scratch4[0], scratch4[1]), fmaxf(scratch4[2], scratch4[3]));
    // a chunk whose largest contribution is below 1e-30 scatters zeros; a non-finite bound (an exploded state) turns the
    // chunk's nodes into NaN, as floating-point accumulation would have
    const bool usable = B > 1e-30f && B < 3e38f;
    to_tile = usable ? FIX_RANGE / B : 0.f;
    from_tile = usable ? B * (1.0f / FIX_RANGE) : (B <= 1e-30f ? 0.f : __builtin_nanf(""));
}
template <class R> __device__ __forceinline__ void lds_add(tile_t* p, R v) {
    __hip_atomic_fetch_add(p, (tile_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <class R> __device__ __forceinline__ void atomic_add(R* p, R v) { unsafeAtomicAdd(p, v); }

// Grid accesses use a wave-uniform base pointer + a 32-bit per-lane BYTE offset, which maps to the
// saddr form of global_load/global_atomic (one VGPR per address instead of a 64-bit pair): with 27
// stencil nodes x several fields per particle the 64-bit form alone overflowed the register budget.
template <class R> __device__ __forceinline__ Vec4<R> gld(const Vec4<R>* base, unsigned cell) {
    return *(const Vec4<R>*)((const char*)base + cell * (unsigned)sizeof(Vec4<R>));
}
#ifndef SMAC_TIMING_PLAIN_GRID_ADDS
#define SMAC_TIMING_PLAIN_GRID_ADDS 0    // 1 (timing experiment only, WRONG results): the grid adds of the hit-list kernels as plain stores - what the atomics cost (profiles/r05_contact_grid.txt)
#endif
template <class R> __device__ __forceinline__ void gatomic(Vec4<R>* base, unsigned cell, int comp, R v) {
#if SMAC_TIMING_PLAIN_GRID_ADDS
    ((R*)((char*)base + cell * (unsigned)sizeof(Vec4<R>)))[comp] = v;
#else
    unsafeAtomicAdd((R*)((char*)base + cell * (unsigned)sizeof(Vec4<R>)) + comp, v);
#endif
}

// Flush of a workgroup's node sums onto a grid field, a node's four words in four NEIGHBOURING LANES of one atomic instruction.  The memory side of the fabric performs
// the adds that reach one 128-byte line one REQUEST after the other (about 10 ns each, whatever the number of lanes in it: tools/microbench/atomic_chain.hip), and a
// launch lasts until the last one is done.  One lane per node and one instruction per word sends a line of 8 nodes FOUR requests per workgroup; this sends it one or
// two (microbenchmark, 64 workgroups on the same 216 nodes: 7.4 -> 3.2 us; 256: 25.8 -> 7.4).  stage / cells: BLOCK * 4 values and BLOCK cell indices in LDS.
// Every thread of the workgroup calls it (two barriers inside) with its node's sum (all zero: nothing to add).
template <class R> __device__ __forceinline__ void flush_nodes_by_lanes(Vec4<R>* field, const Vec4<R>& o, unsigned cell, R* stage, unsigned* cells) {
    const int t = threadIdx.x;
    stage[4 * t + 0] = o.x; stage[4 * t + 1] = o.y; stage[4 * t + 2] = o.z; stage[4 * t + 3] = o.w;
    cells[t] = cell;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = (t & ~63) + 16 * q + ((t & 63) >> 2), c = t & 3;
        const R v = stage[4 * n + c];
        if (v != R(0)) gatomic(field, cells[n], c, v);
    }
    __syncthreads();
}

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
    const size_t po = poff(p);
#pragma unroll
    for (int i = 0; i < cnt; ++i) out[i] = fr[rowbase(c0 + i, Npad) + po];
}
// position rows (smac_math.hpp pos_of): doubles, or 32-bit fixed point in the float slots
template <class R> __device__ __forceinline__ void load_pos(const R* fr, int Npad, int p, typename pos_of<R>::type* out) {
    typedef typename pos_of<R>::type P;
    const size_t po = poff(p);
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = ((const P*)fr)[rowbase(CX + i, Npad) + po];
}
// The 24 adjoint rows k_p2g_grad writes go out non-temporal: measured over 4 + 4 processes (profiles/r02_z_nt_stores.txt) 91.6 -> 89.1 us, against
// +0.4 us in the k_g2p_grad that reads them a substep later.  The same on k_g2p's 15 rows saves it 1.4 us and costs the next k_p2g, which reads them
// 80 us later, 1.2 us: those stay plain.
#ifndef SMAC_NT_STORES
#define SMAC_NT_STORES 1
#endif
template <class T> __device__ __forceinline__ void row_store_nt(T* p, T v) {
#if SMAC_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <class R> __device__ __forceinline__ void store_pos(R* fr, int Npad, int p, int i, typename pos_of<R>::type v) {
    typedef typename pos_of<R>::type P;
    ((P*)fr)[rowoff(CX + i, p, Npad)] = v;
}
template <class R, class P> __device__ __forceinline__ void pos_to(const P* x, R* out) {
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = (R)pos_get(x[i]);
}
template <class R> __device__ __forceinline__ typename pos_of<R>::type pos_mid() {
    typename pos_of<R>::type v;
    pos_set(0.5, v);
    return v;
}

// Addresses of the 27 stencil nodes: block-major global cell = cx[i] + cy[j] + cz[k]; tile-local word
// = tx[i] + ty[j] + tz[k] when the node lies inside the chunk's 6^3 tile (bit i of okx etc.).
// The base is clamped into the grid for ADDRESSING only (the reference would touch memory outside its
// fields for a particle that left the [1.5dx, 1-1.5dx] box; we must not).
struct Nodes {
    int cx[3], cy[3], cz[3];
    int tx[3], ty[3], tz[3];
    int okx, oky, okz;
    __device__ __forceinline__ unsigned cell(int i, int j, int k) const { return (unsigned)(cx[i] + cy[j] + cz[k]); }
    __device__ __forceinline__ bool in_tile(int i, int j, int k) const { return ((okx >> i) & (oky >> j) & (okz >> k) & 1) != 0; }
    __device__ __forceinline__ int tile(int i, int j, int k) const { return tx[i] + ty[j] + tz[k]; }
};
// WIDE: indices and in-tile bits for the particle kernels' tile (PW, PO, PSX, PSY); otherwise for the 6^3 tile of the contact kernels
template <bool WIDE, class R> __device__ __forceinline__ void stencil_at_t(const DevSim<R>& D, const typename pos_of<R>::type* x, Stencil<R>& st, Nodes& nd, int block) {
    constexpr int W_ = WIDE ? PW : TW, O_ = WIDE ? PO : 0, SX_ = WIDE ? PSX : TSX, SY_ = WIDE ? PSY : TSY;
    make_stencil_pos(x, D.n, st);
    const int nb = D.nb;
    const int bz = block % nb, by = (block / nb) % nb, bx = block / (nb * nb);
    const int org[3] = {4 * bx - O_, 4 * by - O_, 4 * bz - O_};
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
        nd.tx[o] = li * SX_; nd.ty[o] = lj * SY_; nd.tz[o] = lk;
        nd.okx |= ((unsigned)li < (unsigned)W_) << o;
        nd.oky |= ((unsigned)lj < (unsigned)W_) << o;
        nd.okz |= ((unsigned)lk < (unsigned)W_) << o;
    }
}
template <class R> __device__ __forceinline__ void stencil_at(const DevSim<R>& D, const typename pos_of<R>::type* x, Stencil<R>& st, Nodes& nd, int block) {
    stencil_at_t<false>(D, x, st, nd, block);
}
template <class R> __device__ __forceinline__ void stencil_at_p(const DevSim<R>& D, const typename pos_of<R>::type* x, Stencil<R>& st, Nodes& nd, int block) {
    stencil_at_t<true>(D, x, st, nd, block);
}

// the particle's material: entry 0 or - MAT2 instantiations, selector byte set - entry 1 of the two-entry table
template <class CT, bool MAT2, class R> __device__ __forceinline__ Material<CT> particle_material(const DevSim<R>& D, int p) {
    Material<CT> mat = {D.mat.ptype, D.mat.model, (CT)D.mat.mu, (CT)D.mat.lam, D.mat.plast, (CT)D.mat.yield_c};
    if (MAT2 && D.mat_id && D.mat_id[D.orig_id[p]]) { mat.mu = (CT)D.mu2; mat.lam = (CT)D.lam2; mat.yield_c = (CT)D.yield_c2; }
    return mat;
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
#if SMAC_PHASE_CLOCK
// [marker][slot]: sums of timestamps (mod 2^64: differences of sums are sums of differences) and hit counts; markers 0-15 backward, 16-31 forward kernels
__device__ unsigned long long smac_phase_sum[48 * 64];
__device__ unsigned long long smac_phase_cnt[48 * 64];
// [kernel: 0 k_contact_hits, 1 k_contact_grad][bin]: EVERY wave's time from entry to exit in bins of 1024 ticks (a launch lasts as long as its slowest wave)
__device__ unsigned long long smac_phase_hist[2 * 64];
#define SMAC_WAVE_T0() const unsigned long long wave_t0_ = __builtin_readcyclecounter(); unsigned long long wave_tm_[6] = {0, 0, 0, 0, 0, 0}, wave_note_ = 0; (void)wave_tm_; (void)wave_note_
#define SMAC_WAVE_HIST(k)                                                                            \
    do {                                                                                             \
        if ((threadIdx.x & 63) == 0) {                                                               \
            const unsigned long long dt_ = (__builtin_readcyclecounter() - wave_t0_) >> 10;          \
            atomicAdd(&smac_phase_hist[(k) * 64 + (int)(dt_ < 63 ? dt_ : 63)], 1ull);                \
        }                                                                                            \
    } while (0)
// the slowest waves of k_contact_grad, one record of 8 words each: {entry-to-exit ticks, workgroup | wave << 16 | band masks of the wave's two hits << 24, hit count,
// ticks from entry to five points of the walk}
__device__ unsigned long long smac_slow[256 * 8];
__device__ unsigned smac_slow_n;
#define SMAC_WAVE_MARK(n) wave_tm_[n] = __builtin_readcyclecounter()
#define SMAC_WAVE_NOTE(x) wave_note_ = (unsigned long long)(x)
#define SMAC_WAVE_SLOW(limit, info, nh_)                                                                \
    do {                                                                                             \
        const unsigned long long end_ = __builtin_readcyclecounter();                                \
        if ((threadIdx.x & 63) == 0 && end_ - wave_t0_ > (limit)) {                                  \
            const unsigned k_ = atomicAdd(&smac_slow_n, 1u);                                         \
            if (k_ < 256u) {                                                                         \
                unsigned long long* r_ = smac_slow + 8 * k_;                                         \
                r_[0] = end_ - wave_t0_; r_[1] = (info) | (wave_note_ << 24); r_[2] = (unsigned long long)(nh_);          \
                for (int q_ = 1; q_ < 6; ++q_) r_[2 + q_] = wave_tm_[q_] - wave_t0_;                     \
            }                                                                                        \
        }                                                                                            \
    } while (0)
#define SMAC_PHASE(i, cond)                                                                          \
    do {                                                                                             \
        if ((threadIdx.x & 63) == 0 && (blockIdx.x & 15) == 3 && (cond)) {                           \
            const int slot_ = (i) * 64 + (int)((blockIdx.x >> 4) & 63);                              \
            atomicAdd(&smac_phase_sum[slot_], (unsigned long long)__builtin_readcyclecounter());     \
            atomicAdd(&smac_phase_cnt[slot_], 1ull);                                                 \
        }                                                                                            \
    } while (0)
#else
#define SMAC_PHASE(i, cond) do { } while (0)
#define SMAC_WAVE_T0() do { } while (0)
#define SMAC_WAVE_HIST(k) do { } while (0)
#define SMAC_WAVE_MARK(n) do { } while (0)
#define SMAC_WAVE_NOTE(x) do { } while (0)
#define SMAC_WAVE_SLOW(...) do { } while (0)
#endif
#define SMAC_CHUNK_PROLOGUE_AT(ci)                            \
    const Chunk ch = D.chunks[ci];                            \
    const int t = threadIdx.x;                                \
    const bool valid = t < ch.count;                          \
    const int p = ch.start + (valid ? t : 0);
// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 names the XCD group).  Chunks are listed in block
// order, so giving XCD g the contiguous range [g*n/8, (g+1)*n/8) keeps spatially adjacent chunks - which re-read the
// same grid records and slabs - behind ONE L2 instead of spreading every grid line over all eight.  Speed only.
__device__ __forceinline__ int xcd_chunk_at(int bid, int nchunks) {
    const int per = (nchunks + 7) >> 3;
    return (bid & 7) * per + (bid >> 3);
}
__device__ __forceinline__ int xcd_chunk(int nchunks) { return xcd_chunk_at((int)blockIdx.x, nchunks); }
// The per-cell grid kernels give every wave one active block (4 per workgroup), in the order of the active list.  (XCD-contiguous ranges, as for
// the chunks, were measured and made k_grid_op / k_reduce_* 0.5 us slower each: profiles/r02_ad_g2p_lds_gather.txt.)
template <class R> __device__ __forceinline__ int active_slot(const DevSim<R>&) { return (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); }
#define SMAC_CHUNK_PROLOGUE                                   \
    const int cid = xcd_chunk(D.nchunks);                     \
    if (cid >= D.nchunks) return;                             \
    SMAC_CHUNK_PROLOGUE_AT(cid)

// ------------------------------------------------------------------------------------------
// Agent-scope ("sc1") accesses: bytes one workgroup hands to ANOTHER workgroup of the same launch (tail reduction below).  A CU's vector L1 is never refreshed
// by other CUs' stores and the eight XCD L2s are not coherent for plain accesses (MI355X_MICROARCH.md, inter-workgroup visibility); the hand-off used here is
// that guide's counter form with write-through stores: EVERY store of the handed-off bytes is an sc1 store, every storing wave drains (s_waitcnt vmcnt(0)), a
// workgroup barrier, ONE wave adds to the counters (agent-scope atomics), the workgroup whose add came last - told by the value the add returned - reads
// the bytes with sc1 loads only.  Float atomics (the drift field) execute at the memory side and are agent-scope by themselves.
// ------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sc1_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00027000);       // raw buffer: byte offsets, 32-bit data format
}
constexpr int AUX_SC1 = 16;
__device__ __forceinline__ Vec4<float> ld_sc1(__amdgpu_buffer_rsrc_t rs, unsigned rec, const Vec4<float>*) {
    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(rec * 16u), 0, AUX_SC1);
    Vec4<float> v = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
    return v;
}
__device__ __forceinline__ Vec4<double> ld_sc1(__amdgpu_buffer_rsrc_t rs, unsigned rec, const Vec4<double>*) {
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(rec * 32u), 0, AUX_SC1), b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(rec * 32u + 16u), 0, AUX_SC1);
    Vec4<double> v = {__hiloint2double((int)a.y, (int)a.x), __hiloint2double((int)a.w, (int)a.z), __hiloint2double((int)b.y, (int)b.x), __hiloint2double((int)b.w, (int)b.z)};
    return v;
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t rs, unsigned rec, const Vec4<float>& v) {
    const u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, rs, (int)(rec * 16u), 0, AUX_SC1);
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t rs, unsigned rec, const Vec4<double>& v) {
    const u32x4 a = {(unsigned)__double2loint(v.x), (unsigned)__double2hiint(v.x), (unsigned)__double2loint(v.y), (unsigned)__double2hiint(v.y)};
    const u32x4 b = {(unsigned)__double2loint(v.z), (unsigned)__double2hiint(v.z), (unsigned)__double2loint(v.w), (unsigned)__double2hiint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(a, rs, (int)(rec * 32u), 0, AUX_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(b, rs, (int)(rec * 32u + 16u), 0, AUX_SC1);
}

// store this chunk's f64 LDS tile (NS scalars, tile[s][word]) to its slab as 16-byte records, coalesced
// (sc1: write-through stores - the slab is read by another workgroup of this launch, see tail_arrive)
template <class R, int NS, class W> __device__ __forceinline__ void tile_store(const DevSim<R>& D, const W* tile, R s0 = R(1), R s123 = R(1), int cid = -1, bool sc1 = false) {
    const unsigned first = (unsigned)(cid >= 0 ? cid : xcd_chunk(D.nchunks)) * TILE_WORDS;
    Vec4<R>* dst = D.slab + first;
    const __amdgpu_buffer_rsrc_t rs = sc1_rsrc(D.slab);
    for (int i = threadIdx.x; i < TILE_WORDS; i += BLOCK) {
        int ti, tj, tk;
        slab_node(i, ti, tj, tk);
        const int w = SMAC_WIDE_TILE ? (ti + PO) * PSX + (tj + PO) * PSY + tk + PO : tile_index(ti, tj, tk);      // the 6^3 core of the wide tile
        Vec4<R> v;
        v.x = (R)tile[w] * s0; v.y = (R)tile[PTILE + w] * s123; v.z = (R)tile[2 * PTILE + w] * s123;
        v.w = NS > 3 ? (R)tile[3 * PTILE + w] * s123 : R(0);
        if (sc1) st_sc1(rs, first + (unsigned)i, v);
        else dst[i] = v;
    }
}
// zero `bytes` (a multiple of 16) of LDS with 16-byte stores
__device__ __forceinline__ void lds_zero16(void* p, int bytes) {
    int4* q = (int4*)p;
    const int4 z = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < bytes / 16; i += BLOCK) q[i] = z;
}
// wide tile: what landed on the one-node shell around the core (particles that crossed a face of their block since the last sort) is added to the dense
// `field`, one global atomic per non-zero word - a few dozen nodes per chunk, pre-reduced over the chunk's particles
template <class R, int NS, class W> __device__ __forceinline__ void tile_flush_shell(const DevSim<R>& D, const W* tile, Vec4<R>* field, int block, R s0, R s123) {
#if SMAC_WIDE_TILE
    const int nb = D.nb;
    const int bz = block % nb, by = (block / nb) % nb, bx = block / (nb * nb);
    // the 296 shell nodes, enumerated: the two x faces (2 x 64), the y faces without them (2 x 6 x 8), the z faces without both (2 x 6 x 6)
    for (int idx = threadIdx.x; idx < 296; idx += BLOCK) {
        int a, b, c;
        if (idx < 128) { a = (idx >> 6) * 7; b = (idx >> 3) & 7; c = idx & 7; }
        else if (idx < 224) { const int r = idx - 128, h = r >= 48 ? 1 : 0, q = r - 48 * h; b = 7 * h; a = 1 + (q >> 3); c = q & 7; }
        else { const int r = idx - 224, h = r >= 36 ? 1 : 0, q = r - 36 * h; c = 7 * h; a = 1 + q / 6; b = 1 + q % 6; }
        const int w = a * PSX + b * PSY + c;
        const W t0 = tile[w], t1 = tile[PTILE + w], t2 = tile[2 * PTILE + w], t3 = NS > 3 ? tile[3 * PTILE + w] : W(0);
        if (t0 == W(0) && t1 == W(0) && t2 == W(0) && t3 == W(0)) continue;
        const int i = 4 * bx - PO + a, j = 4 * by - PO + b, k = 4 * bz - PO + c;
        if ((unsigned)i >= (unsigned)D.n || (unsigned)j >= (unsigned)D.n || (unsigned)k >= (unsigned)D.n) continue;   // (never addressed: bases are clamped)
        const unsigned cell = (unsigned)cell_of(nb, i, j, k);
        if (t0 != W(0)) gatomic(field, cell, 0, (R)t0 * s0);
        if (t1 != W(0)) gatomic(field, cell, 1, (R)t1 * s123);
        if (t2 != W(0)) gatomic(field, cell, 2, (R)t2 * s123);
        if (NS > 3 && t3 != W(0)) gatomic(field, cell, 3, (R)t3 * s123);
    }
#endif
}

// Sum, for one cell of block `b`, every slab record that overlaps it (own block and the blocks at -1
// along each dimension in which the cell's local coordinate is <= 1).
template <class R, bool SC1 = false>
__device__ __forceinline__ void slab_reduce(const DevSim<R>& D, int b, int l, Vec4<R>& acc) {
    // Called by a whole wave for the 64 cells of block b.  The <= 8 source blocks (b and its -1 neighbours) are the same
    // for every lane: lanes 0..7 fetch their chunk ranges in ONE round trip and broadcast them; then up to 4 slab
    // records per source are loaded back to back (predicated), so the loads overlap instead of forming a chain of
    // ~24 dependent L2 round trips per wave (which is what made the grid kernels take 15 us).
    const int nb = D.nb;
    const int bz = b % nb, by = (b / nb) % nb, bx = b / (nb * nb);
    const int lx = l >> 4, ly = (l >> 2) & 3, lz = l & 3;
    const int ex = (lx <= 1) ? 1 : 0, ey = (ly <= 1) ? 1 : 0, ez = (lz <= 1) ? 1 : 0;
    const int lane = threadIdx.x & 63;
    int my_nch = 0, my_start = 0;
    if (lane < 8) {
        const int sx = bx - (lane >> 2), sy = by - ((lane >> 1) & 1), sz = bz - (lane & 1);
        if (sx >= 0 && sy >= 0 && sz >= 0) {
            const int src = (sx * nb + sy) * nb + sz;
            my_nch = D.block_chunks[src];
            my_start = D.block_chunk_start[src];
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int nch = __shfl(my_nch, q, 64), start = __shfl(my_start, q, 64);        // wave-uniform
        if (nch == 0) continue;
        const int dx = q >> 2, dy = (q >> 1) & 1, dz = q & 1;
        const bool mine = dx <= ex && dy <= ey && dz <= ez;
        const int w = mine ? slab_record(q, lx, ly, lz) : 0;
        const Vec4<R>* sl = D.slab + (size_t)start * TILE_WORDS + w;
        const unsigned rec0 = (unsigned)start * TILE_WORDS + (unsigned)w;
        const __amdgpu_buffer_rsrc_t rs = sc1_rsrc(D.slab);
        auto rd = [&](int c) { return SC1 ? ld_sc1(rs, rec0 + (unsigned)c * TILE_WORDS, sl) : sl[(size_t)c * TILE_WORDS]; };
        const Vec4<R> z = {R(0), R(0), R(0), R(0)};
        Vec4<R> v0 = z, v1 = z, v2 = z, v3 = z;
        if (mine) {
            v0 = rd(0);
            if (nch > 1) v1 = rd(1);
            if (nch > 2) v2 = rd(2);
            if (nch > 3) v3 = rd(3);
        }
        acc.x += (v0.x + v1.x) + (v2.x + v3.x); acc.y += (v0.y + v1.y) + (v2.y + v3.y);
        acc.z += (v0.z + v1.z) + (v2.z + v3.z); acc.w += (v0.w + v1.w) + (v2.w + v3.w);
        if (mine)
            for (int c = 4; c < nch; ++c) {
                const Vec4<R> v = rd(c);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
    }
}

// Stage the 6x6x6 node records of `field` around this chunk's block into LDS (gather tile).
// Nodes outside the grid read as zero (they are never addressed: bases are clamped).
template <class R>
__device__ __forceinline__ void gather_tile_load(const DevSim<R>& D, const Vec4<R>* field, int block, Vec4<R>* gt) {
    const int nb = D.nb;
    const int bz = block % nb, by = (block / nb) % nb, bx = block / (nb * nb);
    for (int idx = threadIdx.x; idx < TILE_WORDS; idx += BLOCK) {
        const int li = idx / TSX, lj = (idx / TSY) % TW, lk = idx % TW;
        const int i = 4 * bx + li, j = 4 * by + lj, k = 4 * bz + lk;
        // (pointer select, not "zero record overwritten by a load": that form leaves a 16-byte per-thread temporary which the compiler promotes to LDS)
        const Vec4<R>* src = (i < D.n && j < D.n && k < D.n) ? field + cell_of(nb, i, j, k) : nullptr;
        if (src) gt[idx] = *src;
        else { gt[idx].x = R(0); gt[idx].y = R(0); gt[idx].z = R(0); gt[idx].w = R(0); }
    }
}

// the same for the particle kernels' tile (PW nodes per axis from 4*block - PO, rows of PSY records)
template <class R>
__device__ __forceinline__ void gather_tile_load_p(const DevSim<R>& D, const Vec4<R>* field, int block, Vec4<R>* gt) {
#if SMAC_WIDE_TILE
    // 512 records = two per thread, both loads issued before either is written to LDS (one round trip, as the 216 records of the 6^3 tile were; the
    // loop form waited for the first record before asking for the second: +2.6 us on k_g2p, profiles/r04_x_wide_tiles.txt).  Nodes outside the grid
    // load cell 0 and are replaced by zeros - no branch around the loads.
    static_assert(PW * PW * PW == 2 * BLOCK, "two records per thread");
    const int nb = D.nb;
    const int bz = block % nb, by = (block / nb) % nb, bx = block / (nb * nb);
    Vec4<R> r[2];
    int w[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int idx = (int)threadIdx.x + q * BLOCK;
        const int li = idx >> 6, lj = (idx >> 3) & 7, lk = idx & 7;
        const int i = 4 * bx - PO + li, j = 4 * by - PO + lj, k = 4 * bz - PO + lk;
        w[q] = li * PSX + lj * PSY + lk;
        const bool ok = (unsigned)i < (unsigned)D.n && (unsigned)j < (unsigned)D.n && (unsigned)k < (unsigned)D.n;
        r[q] = gld(field, ok ? (unsigned)cell_of(nb, i, j, k) : 0u);
        if (!ok) { r[q].x = R(0); r[q].y = R(0); r[q].z = R(0); r[q].w = R(0); }
    }
    gt[w[0]] = r[0];
    gt[w[1]] = r[1];
#else
    gather_tile_load(D, field, block, gt);
#endif
}

// zero `nfields` consecutive 4-scalar grid fields on the active blocks
template <class R>
__global__ __launch_bounds__(BLOCK) void k_clear_active(DevSim<R> D, Vec4<R>* base, int nfields) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { *D.nhits = 0; *D.ncand = 0; }   // contact lists are rebuilt by k_p2g
    const int a = active_slot(D);
    if (a >= D.nactive) return;
    const size_t cell = (size_t)D.active[a] * 64 + (threadIdx.x & 63);
    const Vec4<R> z = {R(0), R(0), R(0), R(0)};
    for (int f = 0; f < nfields; ++f) base[(size_t)f * D.G + cell] = z;
}

// next free slot of a device-side list: the active lanes of the wave share ONE atomic on the counter
__device__ __forceinline__ int hit_slot(int* counter) {
    const unsigned long long act = __ballot(1);
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)act) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(act));
    base = __shfl(base, leader, 64);
    return base + __popcll(act & ((1ull << lane) - 1ull));
}

// band test shared by k_contact and k_contact_grad: which primitives see this particle
// This is a FILTER: in f32 it runs in float on a float copy of the table and lets a margin of 2e-5 through; the contact
// kernels repeat the test (collide_* return false outside the band) in the arithmetic that decides.
template <class R> __device__ __forceinline__ void prim_state_R(const DevSim<R>& D, int i, int f, R* s13) {
    const double* ps = D.prim_state + ((size_t)i * D.max_frames + f) * 13;
#pragma unroll
    for (int c = 0; c < 13; ++c) s13[c] = (R)ps[c];
}
// `ps`: optional workgroup copy of the primitive states of frame f already converted to R ([prim][13], k_p2g: 13 f64 -> f32
// conversions per primitive once per workgroup instead of once per particle)
template <class R> __device__ __forceinline__ int contact_mask(const DevSim<R>& D, int f, const typename pos_of<R>::type* xp, const R* ps = nullptr) {
    int mask = 0;
    R x[3];
    pos_to(xp, x);
#pragma unroll
    for (int i = 0; i < MAX_PRIMS; ++i) {
        if (i >= D.P || !D.prim[i].contact) continue;
        R st[13];
        if (ps) {
#pragma unroll
            for (int c = 0; c < 13; ++c) st[c] = ps[13 * i + c];
        } else prim_state_R(D, i, f, st);
        R d = prim_sdf(D.prim[i], st, x);
        if (d <= R(5e-3) + (sizeof(R) == 4 ? R(2e-5) : R(0))) mask |= 1 << i;
    }
    return mask;
}

// Contact is sparse: only particles inside a primitive's 5e-3 band do anything in grid_op_mixed2-4
// (for the others v_tgt == v_tmp and the mixed4 correction is exactly zero).  k_p2g evaluates the band
// test while it has x in registers and builds (a) a compact hit list {particle, mask, block} walked by
// the contact adjoint and (b) the list of chunks holding hits + a per-particle mask, walked by k_contact.
// stand-alone form of the same test (used when the forward grid is restored from a checkpoint)
template <class R>
__global__ __launch_bounds__(BLOCK) void k_contact_mask(DevSim<R> D, int f) {
    SMAC_CHUNK_PROLOGUE
    int cmask = 0;
    if (valid) {
        typename pos_of<R>::type x[3];
        load_pos(frame(D.S, f, D.Npad), D.Npad, p, x);
        cmask = contact_mask(D, f, x);
        if (cmask) {
            Hit h = {p, cmask, ch.block, 0};
            D.hits[hit_slot(D.nhits)] = h;
        }
    }
    if (D.collision_type == CONTACT_PARTICLE && valid) D.pmask[p] = cmask;
}

// tail reduction (defined behind boundary(): the forward reducer applies grid_op's boundary rule)
template <class R> __device__ __forceinline__ void tail_block_fwd(const DevSim<R>& D, int b);
template <class R> __device__ __forceinline__ void tail_block_bwd(const DevSim<R>& D, int b);
template <class R, class FN> __device__ __forceinline__ void tail_arrive(const DevSim<R>& D, int block, int* lds_word, FN reduce);

// Grid checkpoint: the three value fields of the active blocks, packed [active slot][field][64 cells].
// Saved after the forward substep's contact pass, restored (with the adjoint fields zeroed) at the start
// of substep_grad instead of recomputing compute_F_tmp/svd/p2g/grid_op (mpm_simulator.py:352-359).
// hit_ck / nhit_ck (optional): the frame's contact hit list travels with the checkpoint, so the backward pass
// does not have to repeat the band test over all particles (k_contact_mask)
// (`nblocks` workgroups share the work: the whole launch for k_grid_save, the first D.save_blocks workgroups of k_g2p<R, true>)
template <class R>
__device__ __forceinline__ void grid_save_block(const DevSim<R>& D, int nblocks, Vec4<R>* ck, Hit* hit_ck, int* nhit_ck, int hit_cap) {
    if (hit_ck) {
        int nh = *D.nhits;
        bool fits = true;
        if (nh > hit_cap) {                  // more particles in contact bands than a checkpoint slot holds: never truncated silently - the host is
            if (blockIdx.x == 0 && threadIdx.x == 0) D.drift_flag[1] = 1;      // told (second flag word) and substep_grad repeats the band test instead
            nh = 0;
            fits = false;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            *nhit_ck = nh;
            if (D.save_nhits_host) *D.save_nhits_host = fits ? nh : -1;  // (pinned host memory: substep_grad skips the contact adjoint of a frame without hits)
        }
        for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nh; i += nblocks * BLOCK) hit_ck[i] = D.hits[i];
    }
    const int a = active_slot(D);
    if (a >= D.nactive) return;
    const int l = threadIdx.x & 63;
    const int b = D.active[a];
    const size_t cell = (size_t)b * 64 + l;
    Vec4<R>* dst = ck + (size_t)a * CK_WORDS + l;
    const Vec4<R> in = D.vin[cell];
    bool empty = false;
    if (D.ck_flags) {                                                    // (a wave = a block)
        empty = __ballot(in.x != R(0) || in.y != R(0) || in.z != R(0) || in.w != R(0)) == 0ull;
        if (l == 0) D.ck_flags[a] = empty ? 1 : 0;
    }
    if (!empty) { dst[0] = in; dst[64] = D.vout[cell]; }                 // (no mass: grid_op left v_mixed = v_out = 0 there, {m,p} is zero already: nothing is filed)
    // ({m, p} stays as it is: the next substep's grid_op overwrites it; what P2G adds with global atomics goes to D.vdrift)
    // Tail reduction: the P2G of the NEXT substep rides in this launch (k_g2p_p2g) and its last arriver overwrites {m,p} and v_out of the block.  This wave is
    // one of the block's readers: it arrives like a chunk (DevSim::tail_extra) once its loads have returned - and, should every chunk have arrived before it,
    // it is the one that reduces.
    if (SMAC_TAIL_BUILD && D.tail_on && D.tail_extra && D.tail_expect[b] > 0) {             // (wave-uniform)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int old = 0;
        if (l == 0) old = __hip_atomic_fetch_add(D.tail_cnt + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old + 1 == D.tail_expect[b] + D.tail_extra) tail_block_fwd(D, b);
    }
}
template <class R>
__global__ __launch_bounds__(BLOCK) void k_grid_save(DevSim<R> D, Vec4<R>* ck, Hit* hit_ck, int* nhit_ck, int hit_cap) {
    grid_save_block(D, (int)gridDim.x, ck, hit_ck, nhit_ck, hit_cap);
}
template <class R>
__global__ __launch_bounds__(BLOCK) void k_grid_restore(DevSim<R> D, const Vec4<R>* ck, const Hit* hit_ck, const int* nhit_ck, int zero_all) {
    if (hit_ck) {
        const int nh = *nhit_ck;
        if (blockIdx.x == 0 && threadIdx.x == 0) { *D.nhits = nh; *D.ncand = 0; }
        for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nh; i += gridDim.x * BLOCK) D.hits[i] = hit_ck[i];
    } else if (blockIdx.x == 0 && threadIdx.x == 0) { *D.nhits = 0; *D.ncand = 0; }
    const int a = active_slot(D);
    if (a >= D.nactive) return;
    const int l = threadIdx.x & 63;
    const size_t cell = (size_t)D.active[a] * 64 + l;
    const Vec4<R>* src = ck + (size_t)a * CK_WORDS + l;
    const Vec4<R> z = {R(0), R(0), R(0), R(0)};
    if (D.ck_flags && D.ck_flags[a]) { D.vin[cell] = z; D.vout[cell] = z; }        // a block that held no mass: nothing was filed
    else { D.vin[cell] = src[0]; D.vout[cell] = src[64]; }
    D.aout[cell] = z;                                // g2p.grad's drifted lanes add to it
    if (zero_all == 1) D.ain[cell] = z;                    // (the fused backward grid pass writes every grid_v_in.grad and never reads grid_v_mixed.grad;
    if (zero_all) D.amix[cell] = z;                        //  2: grid_v_in.grad of the substep before is still to be read - the slab pieces' fused backward launch)
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// PCON: collision_type 1 (penalty contact inside p2g) - a separate instantiation, so that the benchmarked forecast-contact
// kernel does not carry the f64 penalty chain in its register budget
// MAT2: the two-entry material table (a separate instantiation: the benchmarked one-material kernel keeps its registers)
// P2G of one chunk (every thread of the workgroup; two barriers).  `tile_raw` is the zeroed scatter tile (zeros visible at the latest by the barrier inside
// tile_scale), `ps_wg` the workgroup's copy of the primitive states of frame f, `cid` the chunk (its slab).  FROM_REGS: x, v, C of frame f come in
// registers (`xr`, `vr`, `Cr`: the G2P of the substep before has just produced them, k_g2p_p2g) instead of from the frame's rows; hits go to `hits` / `nhits`.
template <class R, bool STORE_F, bool PCON, bool MAT2, bool FROM_REGS>
__device__ __forceinline__ void p2g_body(const DevSim<R>& D, int f, const Chunk& ch, int cid, int t, bool valid, int p, double* tile_raw, R* smax, const R* ps_wg,
                                         const typename pos_of<R>::type* xr, const R* vr, const R* Cr, struct Hit* hits, int* nhits) {
    typedef typename ScatterTile<R>::word W;
    W* const tile = (W*)tile_raw;
    double* const tile64 = tile_raw;
    const bool sparse = sizeof(R) == 4 && ch.count <= SPARSE_MAX;       // workgroup-uniform: f64 words instead of fixed point
    int cmask = 0;
    SMAC_PHASE(16, valid);
    typedef typename pos_of<R>::type PX;
    PX x[3] = {pos_mid<R>(), pos_mid<R>(), pos_mid<R>()};
    R pv[3] = {R(0), R(0), R(0)}, aff[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) aff[i] = R(0);
    R bound = R(0);                       // no single scattered momentum component of this particle exceeds it
    if (valid) {
        R Et[9], En[9], stress[9];
        const R* Sf = frame(D.S, f, D.Npad);
        R v[3], C[9], E[9];
        if (FROM_REGS) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { x[i] = xr[i]; v[i] = vr[i]; }
#pragma unroll
            for (int i = 0; i < 9; ++i) C[i] = Cr[i];
        } else {
            load_pos(Sf, D.Npad, p, x);
            load_vec(Sf, CV, 3, D.Npad, p, v);
            load_vec(Sf, CC, 9, D.Npad, p, C);
        }
        load_vec(Sf, CF, 9, D.Npad, p, E);
        int cloth_face = -1;
        if (PCON && D.cloth.present) {                             // cloth primitive, penalty contact: the contact face was searched before the substep
            const size_t at = (size_t)f * D.cloth.n_ids + D.orig_id[p];
            cloth_face = D.cloth.contact_id[at];
            if (cloth_face >= 0) {
                cmask = 1 | ((D.cloth.penetration[at] == 1 ? 1 : 0) << 1);
                Hit h = {p, cmask, ch.block, cloth_face};
                hits[hit_slot(nhits)] = h;
            }
        } else if (D.any_contact && D.collision_type != CONTACT_GRID && !D.cloth.present) {   // contact band test (x is at hand): build the sparse contact lists
            cmask = contact_mask(D, f, x, (const R*)ps_wg);
            if (cmask) {
                Hit h = {p, cmask, ch.block, 0};
                hits[hit_slot(nhits)] = h;
            }
        }
        {
            typedef typename const_t<R>::type CT;
            CT Cc[9], Ec[9], Etc[9], Enc[9], sc[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) { Cc[i] = (CT)C[i]; Ec[i] = (CT)E[i]; }
            f_tmp(Cc, Ec, (CT)D.dt, Etc);
            const Material<CT> mat = particle_material<CT, MAT2>(D, p);
            ConstState<CT> cs;
            constitutive_fwd(mat, Etc, Enc, sc, cs);
#pragma unroll
            for (int i = 0; i < 9; ++i) { Et[i] = (R)Etc[i]; En[i] = (R)Enc[i]; stress[i] = (R)sc[i]; }
        }
        if (STORE_F) {
            R* Sn = frame(D.S, f + 1, D.Npad);
#pragma unroll
            for (int i = 0; i < 9; ++i) Sn[rowoff(CF + i, p, D.Npad)] = En[i];     // :250
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) aff[i] = D.stress_scale * stress[i] + D.p_mass * C[i];   // :247-248
        pv[0] = D.p_mass * v[0]; pv[1] = D.p_mass * v[1]; pv[2] = D.p_mass * v[2];
        if (PCON && cmask && D.cloth.present) {                                           // soft_cloth p2g :208-213 (collide_particle of the sheet)
            const ClothDev& Cl = D.cloth;
            const double sc = Cl.par.scale;
            const int* vid = Cl.faces + 3 * cloth_face;
            double xv[3][3], vv[3][3];
            for (int i = 0; i < 3; ++i)
                for (int c = 0; c < 3; ++c) {
                    xv[i][c] = Cl.pos[((size_t)f * Cl.V + vid[i]) * 3 + c];
                    vv[i][c] = Cl.vel[((size_t)f * Cl.V + vid[i]) * 3 + c];
                }
            const double pp[3] = {sc * pos_get(x[0]), sc * pos_get(x[1]), sc * pos_get(x[2])}, pvv[3] = {sc * (double)v[0], sc * (double)v[1], sc * (double)v[2]};
            double imp[3], cf[3], wb[3];
            if (cloth_collide_particle<double>(Cl.par, xv, vv, pp, pvv, D.dt64, (cmask >> 1) & 1, imp, cf, wb)) {
                const double s3 = 1.0 / (sc * sc * sc);                                   // physical momentum -> unit domain
                for (int c = 0; c < 3; ++c) pv[c] += (R)(imp[c] * s3);
                for (int i = 0; i < 3; ++i)
                    for (int c = 0; c < 3; ++c) atomic_add(Cl.ext_f + (size_t)vid[i] * 3 + c, cf[c] * wb[i]);   // :227-229
            }
        } else if (PCON && cmask) {                                                       // :203-206 penalty contact impulse
            // in f64 whatever R is: the impulse is proportional to the penetration depth c = dist - 5e-3, a small difference
            // of the position and the table (a float x would put 3e-5 on it)
            const double x64[3] = {pos_get(x[0]), pos_get(x[1]), pos_get(x[2])}, v64[3] = {(double)v[0], (double)v[1], (double)v[2]};
#pragma unroll 1
            for (int i = 0; i < D.P; ++i) {
                if (!((cmask >> i) & 1)) continue;
                const double* ps = D.prim_state + ((size_t)i * D.max_frames + f) * 13;
                double s13[13], imp[3], ext[6];
                for (int c = 0; c < 13; ++c) s13[c] = ps[c];
                if (collide_particle(D.prim64[i], s13, x64, v64, D.dt64, imp, ext)) {
                    for (int c = 0; c < 3; ++c) pv[c] += (R)imp[c];
                    for (int c = 0; c < 6; ++c) atomic_add(D.ext_f + i * 6 + c, ext[c]);   // sparse: a few thousand particles
                }
            }
        }
        if (D.n_control > 0) {                                                            // :209-213
            int ci = D.control_idx[D.orig_id[p]];
            if (ci >= 0)
                for (int d = 0; d < 3; ++d) pv[d] += R(6e-4) * D.action[3 * ci + d] * D.dt;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)      // |w (pv + aff dpos)|
            bound = maxc(bound, R(W_MAX) * abs_(pv[c]) + R(WD_MAX) * D.dx * (abs_(aff[3 * c]) + abs_(aff[3 * c + 1]) + abs_(aff[3 * c + 2])));
    }
    // tile units: momentum by the chunk's bound, mass by p_mass (w <= 1); one barrier (all threads)
    R to_tile, from_tile;
    SMAC_PHASE(17, valid);                     // rows loaded, SVD + stress done
    tile_scale<R>(bound, smax, to_tile, from_tile);
    SMAC_PHASE(18, valid);                     // barrier
    const R mass_unit = sizeof(R) == 4 ? R(FIX_RANGE / W_MAX) : D.p_mass;    // w <= W_MAX
    if (valid) {
        Stencil<R> st;
        Nodes nd;
        stencil_at_p(D, x, st, nd, ch.block);
        // the scattered momentum is affine in the node offset: mom(i,j,k) = m0 + i a0 + j a1 + k a2 with
        // a_d = dx * affine[:, d] and m0 = pv - affine (fx dx): three adds per node instead of nine FMAs
        R m0[3], a0[3], a1[3], a2[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a0[c] = D.dx * aff[3 * c]; a1[c] = D.dx * aff[3 * c + 1]; a2[c] = D.dx * aff[3 * c + 2];
            m0[c] = pv[c] - (a0[c] * st.fx[0] + a1[c] * st.fx[1] + a2[c] * st.fx[2]);
        }
        // Every stencil of the wave inside the tile: straight-line LDS atomics.  Otherwise (a lane drifted out of its
        // block since the last sort) the same unrolled code with a per-node choice LDS tile / global atomic - ONE pass
        // for all lanes, where a separate slow loop would make the whole wave run both paths.
        const bool wave_in = __all((nd.okx & nd.oky & nd.okz) == 7);
        if (sparse) {
#pragma unroll 1
            for (int n = 0; n < 27; ++n) {
                const int i = n / 9, j = (n / 3) % 3, k = n % 3;
                const R w = (i == 0 ? st.w[0][0] : (i == 1 ? st.w[1][0] : st.w[2][0])) * (j == 0 ? st.w[0][1] : (j == 1 ? st.w[1][1] : st.w[2][1])) *
                            (k == 0 ? st.w[0][2] : (k == 1 ? st.w[1][2] : st.w[2][2]));
                const R val[3] = {w * (m0[0] + R(i) * a0[0] + R(j) * a1[0] + R(k) * a2[0]), w * (m0[1] + R(i) * a0[1] + R(j) * a1[1] + R(k) * a2[1]),
                                  w * (m0[2] + R(i) * a0[2] + R(j) * a1[2] + R(k) * a2[2])};
                const int tix = (i == 0 ? nd.tx[0] : (i == 1 ? nd.tx[1] : nd.tx[2])) + (j == 0 ? nd.ty[0] : (j == 1 ? nd.ty[1] : nd.ty[2])) +
                                (k == 0 ? nd.tz[0] : (k == 1 ? nd.tz[1] : nd.tz[2]));
                if (((nd.okx >> i) & (nd.oky >> j) & (nd.okz >> k) & 1) != 0) {
                    tile_add(tile64 + tix, w * D.p_mass);
                    for (int c = 0; c < 3; ++c) tile_add(tile64 + tix + (1 + c) * PTILE, val[c]);
                } else {
                    const unsigned cell = (unsigned)((i == 0 ? nd.cx[0] : (i == 1 ? nd.cx[1] : nd.cx[2])) + (j == 0 ? nd.cy[0] : (j == 1 ? nd.cy[1] : nd.cy[2])) +
                                                     (k == 0 ? nd.cz[0] : (k == 1 ? nd.cz[1] : nd.cz[2])));
                    gatomic(D.vdrift, cell, 0, w * D.p_mass);
                    for (int c = 0; c < 3; ++c) gatomic(D.vdrift, cell, 1 + c, val[c]);
                }
            }
        } else if (wave_in) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                R mi[3] = {m0[0] + R(i) * a0[0], m0[1] + R(i) * a0[1], m0[2] + R(i) * a0[2]};
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const R wij = st.w[i][0] * st.w[j][1];
                    R mj[3] = {mi[0] + R(j) * a1[0], mi[1] + R(j) * a1[1], mi[2] + R(j) * a1[2]};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const R w = wij * st.w[k][2];
                        const R ws = w * to_tile;
                        W* tp = tile + nd.tile(i, j, k);
                        tile_add(tp, w * mass_unit);                                       // :262
#pragma unroll
                        for (int c = 0; c < 3; ++c) tile_add(tp + (1 + c) * PTILE, ws * (mj[c] + R(k) * a2[c]));   // :261
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                R mi[3] = {m0[0] + R(i) * a0[0], m0[1] + R(i) * a0[1], m0[2] + R(i) * a0[2]};
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const R wij = st.w[i][0] * st.w[j][1];
                    R mj[3] = {mi[0] + R(j) * a1[0], mi[1] + R(j) * a1[1], mi[2] + R(j) * a1[2]};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const R w = wij * st.w[k][2];
                        const R val[3] = {w * (mj[0] + R(k) * a2[0]), w * (mj[1] + R(k) * a2[1]), w * (mj[2] + R(k) * a2[2])};
                        if (nd.in_tile(i, j, k)) {
                            W* tp = tile + nd.tile(i, j, k);
                            tile_add(tp, w * mass_unit);
#pragma unroll
                            for (int c = 0; c < 3; ++c) tile_add(tp + (1 + c) * PTILE, val[c] * to_tile);
                        } else {
                            const unsigned cell = nd.cell(i, j, k);
                            gatomic(D.vdrift, cell, 0, w * D.p_mass);
#pragma unroll
                            for (int c = 0; c < 3; ++c) gatomic(D.vdrift, cell, 1 + c, val[c]);
                        }
                    }
                }
            }
        }
    }
    if (PCON && D.any_contact && valid) D.pmask[p] = cmask;   // read back by p2g.grad
    SMAC_PHASE(19, valid);                     // scatter issued
    __syncthreads();
    SMAC_PHASE(20, valid);
    const bool tail = SMAC_TAIL_BUILD && D.tail_on != 0;               // (launch-uniform)
    if (sparse) { tile_store<R, 4>(D, tile64, R(1), R(1), cid, tail); tile_flush_shell<R, 4>(D, tile64, D.vdrift, ch.block, R(1), R(1)); }
    else {
        const R s_m = sizeof(R) == 4 ? D.p_mass * R(W_MAX / FIX_RANGE) : R(1);
        tile_store<R, 4>(D, tile, s_m, from_tile, cid, tail);
        tile_flush_shell<R, 4>(D, tile, D.vdrift, ch.block, s_m, from_tile);
    }
    SMAC_PHASE(21, valid);
    // tail reduction: this chunk has delivered to the 27 blocks around its own; for each of them whose count it completes, it sums the block's slab records and
    // the drift field, applies grid_op (mpm_simulator.py:283-297, 396-404) and writes {m,p} and v_out - k_grid_op's work, without the launch
    if (tail) tail_arrive(D, ch.block, (int*)smax, [&](int B) { tail_block_fwd(D, B); });
}

template <class R, bool STORE_F, bool PCON, bool MAT2 = false>
__global__ __launch_bounds__(BLOCK, (sizeof(R) == 4 ? SMAC_OCC_P2G : 2)) void k_p2g(DevSim<R> D, int f) {
    typedef typename ScatterTile<R>::word W;
    __shared__ __attribute__((aligned(16))) double tile_raw[4 * PTILE];
    __shared__ R smax[4];
    __shared__ R ps_wg[MAX_PRIMS * 13];
    SMAC_PHASE(22, true);                // (entry, every wave)
    if (D.any_contact) {                 // (uniform) primitive states of this frame, converted once per workgroup for the band test
        if (threadIdx.x < D.P * 13)
            ps_wg[threadIdx.x] = (R)D.prim_state[((size_t)(threadIdx.x / 13) * D.max_frames + f) * 13 + threadIdx.x % 13];
        __syncthreads();
    }
    SMAC_CHUNK_PROLOGUE
    SMAC_PHASE(23, ch.count >= 0);       // (descriptor in, every wave)
    lds_zero16(tile_raw, 4 * PTILE * (int)((sizeof(R) == 4 && ch.count <= SPARSE_MAX) ? sizeof(double) : sizeof(W)));     // (made visible by the barrier inside tile_scale)
    p2g_body<R, STORE_F, PCON, MAT2, false>(D, f, ch, cid, t, valid, p, tile_raw, smax, ps_wg, nullptr, (const R*)nullptr, (const R*)nullptr, D.hits, D.nhits);
}

// boundary_condition :268-281 on a velocity; returns mask bits of the components that were zeroed
template <class R> __device__ __forceinline__ int boundary(const DevSim<R>& D, int i, int j, int k, R* v) {
    const int I[3] = {i, j, k};
    int mask = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const bool wall_lo = !(d == 0 && (D.open_x & 1)), wall_hi = !(d == 0 && (D.open_x & 2));
        if (wall_lo && I[d] < 3 && v[d] < R(0)) { v[d] = R(0); mask |= 1 << d; }
        if (wall_hi && I[d] > D.n - 3 && v[d] > R(0)) { v[d] = R(0); mask |= 1 << d; }
    }
    if (D.sticky && j < 3) { v[0] = v[1] = v[2] = R(0); mask = 7; }                   // :278-279
    return mask;
}

// Adjoint of grid_op :283-297 / grid_op_mixed1 :396-404 at ONE node (no grid-node contact): `in` = forward {m, p}, `g` = adjoint of
// the node's output velocity (zeroed here where the boundary condition zeroed the velocity); returns {grid_m.grad, grid_v_in.grad}.
// Linear in g, which lets the contact adjoint push each of its sparse contributions through it separately (k_contact_grad).
template <class R> __device__ __forceinline__ Vec4<R> grid_op_node_adjoint(const DevSim<R>& D, const Vec4<R>& in, int i, int j, int k, R* g) {
    const Vec4<R> zero4 = {R(0), R(0), R(0), R(0)};
    const R m = in.x;
    if (!(m > D.m_eps)) return zero4;
    const R inv = R(1) / m;
    R v[3] = {inv * in.y + D.dt * D.g[0], inv * in.z + D.dt * D.g[1], inv * in.w + D.dt * D.g[2]};
    const int mask = boundary(D, i, j, k, v);
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (mask & (1 << d)) g[d] = R(0);
    const R gm = -(in.y * g[0] + in.z * g[1] + in.w * g[2]);
    const Vec4<R> o = {gm * inv * inv, g[0] * inv, g[1] * inv, g[2] * inv};
    return o;
}
// grid coordinates of a block-major cell index
__device__ __forceinline__ void cell_ijk(int nb, unsigned cell, int& i, int& j, int& k) {
    const int b = (int)(cell >> 6), l = (int)(cell & 63u);
    i = 4 * (b / (nb * nb)) + (l >> 4);
    j = 4 * ((b / nb) % nb) + ((l >> 2) & 3);
    k = 4 * (b % nb) + (l & 3);
}

// grid_v_mixed at one node from its {m, p} (grid_op_mixed1 :399-403; the same expressions as k_grid_op).  The field itself is only stored for the slab phases:
// the contact kernels - the only readers - have the node's {m, p} in hand anyway, so the whole-substep path neither writes, files nor restores it (round 4:
// 16 MB less per substep pair in the grid passes, which run at the bandwidth of their 16-byte accesses).
template <class R> __device__ __forceinline__ Vec4<R> grid_v_mixed_at(const DevSim<R>& D, const Vec4<R>& in, unsigned cell) {
    Vec4<R> o = {R(0), R(0), R(0), R(0)};
    if (!(in.x > D.m_eps)) return o;
    const R inv = R(1) / in.x;
    R v[3] = {inv * in.y + D.dt * D.g[0], inv * in.z + D.dt * D.g[1], inv * in.w + D.dt * D.g[2]};
    int i, j, k;
    cell_ijk(D.nb, cell, i, j, k);
    boundary(D, i, j, k, v);
    o.x = v[0]; o.y = v[1]; o.z = v[2];
    return o;
}

// ------------------------------------------------------------------------------------------
// Tail reduction (round 5).  What completed a scatter so far was a launch of its own: k_grid_op summed, per node, the <= 8 slabs that overlap it (42 MB moved
// for 6 MB of result) and applied grid_op; the backward pass had the same in k_reduce_grid_grad_ahead.  Now the scatter's own launch finishes the job:
//   * every chunk, after its slab (sc1 stores) and its drift-field atomics have LEFT (s_waitcnt vmcnt(0) in every wave, then the workgroup barrier), adds 1 to
//     the arrival counter of each of the 27 blocks around its own - one wave instruction, 27 lanes, agent-scope atomics that return the old value;
//   * a block's count is complete when it reaches the number of chunks in those 27 blocks (tail_expect, made at the re-sort) [+ 1: the checkpoint-save wave
//     that still reads the block's previous contents in the same launch];  the workgroup whose add completed it knows from the returned value and reduces the
//     block - one wave per block, slab records and the drift field read with sc1 loads - and resets the counter;
//   * no workgroup ever WAITS: no residency assumption, no ordering assumption, nothing to deadlock.
// The readers of what the reducer overwrites are exactly the contributors: a chunk gathers its tile from the blocks it scatters to, and it arrives after both.
// ------------------------------------------------------------------------------------------
template <class R, class FN> __device__ __forceinline__ void tail_arrive(const DevSim<R>& D, int block, int* lds_word, FN reduce) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // every storing wave: its sc1 stores and atomics are out
    __syncthreads();
    const int nb = D.nb;
    const int bz = block % nb, by = (block / nb) % nb, bx = block / (nb * nb);
    if (threadIdx.x < 64) {                                               // ONE wave signals for the workgroup
        const int t = threadIdx.x;
        bool last = false;
        if (t < 27) {
            const int X = bx + t / 9 - 1, Y = by + (t / 3) % 3 - 1, Z = bz + t % 3 - 1;
            if ((unsigned)X < (unsigned)nb && (unsigned)Y < (unsigned)nb && (unsigned)Z < (unsigned)nb) {
                const int B = (X * nb + Y) * nb + Z;
                const int old = __hip_atomic_fetch_add(D.tail_cnt + B, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = old + 1 == D.tail_expect[B] + D.tail_extra;
            }
        }
        const unsigned long long bal = __ballot(last);
        if (t == 0) *lds_word = (int)(unsigned)bal;
    }
    __syncthreads();                                                      // the other waves load only behind the barrier the signalling wave joins after its adds returned
    unsigned mask = (unsigned)*lds_word;
    const int wave = (int)(threadIdx.x >> 6);
    int k = 0;
    while (mask) {                                                        // (workgroup-uniform) the completed blocks, dealt to the four waves
        const int q = __ffs((int)mask) - 1;
        mask &= mask - 1u;
        if ((k++ & 3) == wave) reduce(((bx + q / 9 - 1) * nb + (by + (q / 3) % 3 - 1)) * nb + (bz + q % 3 - 1));
    }
}

// one wave, lane = cell of block b: {m,p} = slab records + drift field (completes P2G), grid_op :283-297 / grid_op_mixed1 :396-404 with the boundary rule
// (k_grid_op's phase 0 without grid-node contact), counter back to zero
template <class R> __device__ __forceinline__ void tail_block_fwd(const DevSim<R>& D, int b) {
    const int l = (int)(threadIdx.x & 63);
    const unsigned cell = (unsigned)b * 64u + (unsigned)l;
    const Vec4<R> z = {R(0), R(0), R(0), R(0)};
    Vec4<R> acc = ld_sc1(sc1_rsrc(D.vdrift), cell, D.vdrift);
    if (acc.x != R(0) || acc.y != R(0) || acc.z != R(0) || acc.w != R(0)) D.vdrift[cell] = z;      // consumed: the field is all zero again
    slab_reduce<R, true>(D, b, l, acc);
    D.vin[cell] = acc;
    const int nb = D.nb;
    const int i = 4 * (b / (nb * nb)) + (l >> 4), j = 4 * ((b / nb) % nb) + ((l >> 2) & 3), k = 4 * (b % nb) + (l & 3);
    Vec4<R> o = z;
    if (acc.x > D.m_eps) {
        const R inv = R(1) / acc.x;
        R v[3] = {inv * acc.y + D.dt * D.g[0], inv * acc.z + D.dt * D.g[1], inv * acc.w + D.dt * D.g[2]};
        boundary(D, i, j, k, v);
        o.x = v[0]; o.y = v[1]; o.z = v[2];
    }
    D.vout[cell] = o;
    if (l == 0) __hip_atomic_store(D.tail_cnt + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// backward: grid_v_out.grad = slab records + what drifted lanes / tile shells added to the field itself (completes g2p.grad's scatter), then the adjoint of
// grid_op / grid_op_mixed1 at the node -> {grid_m.grad, grid_v_in.grad} (k_reduce_grid_grad's work).  The block's forward {m,p} were restored by an earlier
// launch (plain loads); D.ck_flags: the block held no mass in this frame - nothing was scattered to it, nothing will gather from it.
template <class R> __device__ __forceinline__ void tail_block_bwd(const DevSim<R>& D, int b) {
    const int l = (int)(threadIdx.x & 63);
    const unsigned cell = (unsigned)b * 64u + (unsigned)l;
    if (!(D.ck_flags && D.ck_flags[D.block_slot[b]])) {
        Vec4<R> acc = ld_sc1(sc1_rsrc(D.aout), cell, D.aout);
        const Vec4<R> in = D.vin[cell];
        slab_reduce<R, true>(D, b, l, acc);
        if (D.any_contact) D.aout[cell] = acc;                           // the contact adjoint gathers grid_v_out.grad at the nodes of its hits
        const int nb = D.nb;
        const int i = 4 * (b / (nb * nb)) + (l >> 4), j = 4 * ((b / nb) % nb) + ((l >> 2) & 3), k = 4 * (b % nb) + (l & 3);
        R g[3] = {acc.x, acc.y, acc.z};
        D.ain[cell] = grid_op_node_adjoint(D, in, i, j, k, g);
    }
    if (l == 0) __hip_atomic_store(D.tail_cnt + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one thread per cell of an active block; returns false past the end
template <class R> __device__ __forceinline__ bool active_cell(const DevSim<R>& D, int& b, int& l, size_t& cell, int& i, int& j, int& k) {
    const int a = active_slot(D);
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

__device__ __forceinline__ void prim_fk_step(double* state, int f, double dt);
// slab reduction (completes P2G) fused with grid_op :283-297 / grid_op_mixed1 :396-404
// phase 0: both; 1: slab reduction only (multi-GPU: the halo planes of {m,p} are summed across slabs next);
// 2: normalisation only
// HALO: the slab loop's two pieces pack / add the shared planes themselves (DevSim::halo_hs) - an instantiation of its own, the single-GPU kernel carries none of it
template <class R, bool GRIDC, bool HALO = false>
__global__ __launch_bounds__(BLOCK) void k_grid_op(DevSim<R> D, int phase) {
    if (D.fk_ride > 0 && blockIdx.x == gridDim.x - 1) {       // forward_kinematics to frame cur_frame + 1 rides here when this substep's G2P launch carries the next
        if ((int)threadIdx.x < D.fk_ride) prim_fk_step(D.prim_state + threadIdx.x * D.fk_stride, D.cur_frame, D.dt64);      // substep's P2G, whose band test reads those states
        return;
    }
    if (D.zero_next_hits && blockIdx.x == 0 && threadIdx.x == 0) *D.nhits_next = 0;
    int b, l, i, j, k;
    size_t cell;
    if (!active_cell(D, b, l, cell, i, j, k)) return;
    Vec4<R> acc = *(phase != 2 ? D.vdrift + cell : D.vin + cell);                       // (one load through a selected pointer: what P2G added with global atomics, or {m,p})
    if (phase != 2) {
        if (acc.x != R(0) || acc.y != R(0) || acc.z != R(0) || acc.w != R(0)) {
            const Vec4<R> z0 = {R(0), R(0), R(0), R(0)};
            D.vdrift[cell] = z0;                                                        // consumed: the field is all zero again
        }
        slab_reduce(D, b, l, acc);
        D.vin[cell] = acc;
        if (HALO && phase == 1 && D.halo_hs.count) {                                    // pack: this rank's partial sums on the shared planes
            const size_t total = (size_t)D.halo_np * D.n * D.n;
            for (int s = 0; s < D.halo_hs.count; ++s) {
                const int pi = i - D.halo_hs.plane0[s];
                if ((unsigned)pi < (unsigned)D.halo_np) D.halo_send[(size_t)D.halo_hs.slot[s] * total + ((size_t)pi * D.n + j) * D.n + k] = acc;
            }
        }
    } else if (HALO && D.halo_hs.count) {                                               // unpack-add: the neighbours' partial sums
        const size_t total = (size_t)D.halo_np * D.n * D.n;
        bool got = false;
        for (int s = 0; s < D.halo_hs.count; ++s) {
            const int pi = i - D.halo_hs.plane0[s];
            if ((unsigned)pi < (unsigned)D.halo_np) {
                const Vec4<R> a = D.halo_recv[(size_t)D.halo_hs.slot[s] * total + ((size_t)pi * D.n + j) * D.n + k];
                acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
                got = true;
            }
        }
        if (got) D.vin[cell] = acc;                                                     // (the checkpoint, the contact kernels and the backward pass read the totals)
    }
    if (phase == 1) return;
    const Vec4<R> z = {R(0), R(0), R(0), R(0)};
    const R m = acc.x;
    if (!(m > D.m_eps)) {                                                             // :286 / :399: no velocity on this node
        if (D.collision_type == CONTACT_MIXED && D.keep_vmix) D.vmix[cell] = z;        // (written, so the fields need no clear)
        D.vout[cell] = z;
        return;
    }
    const R inv = R(1) / m;
    R v[3] = {inv * acc.y + D.dt * D.g[0], inv * acc.z + D.dt * D.g[1], inv * acc.w + D.dt * D.g[2]};   // :287-288
    if (GRIDC) {                                                                       // :290-294 grid-node contact (collision_type 0)
        const R pos[3] = {R(i) * D.dx, R(j) * D.dx, R(k) * D.dx};
#pragma unroll 1
        for (int q = 0; q < D.P; ++q) {
            if (!D.prim[q].contact) continue;
            R s13[13], ext[6];
            prim_state_R(D, q, D.cur_frame, s13);
            if (collide_grid(D.prim[q], s13, pos, v, m, D.dt, ext))
                for (int c = 0; c < 6; ++c) atomic_add(D.ext_f + q * 6 + c, (double)ext[c]);
        }
    }
    boundary(D, i, j, k, v);
    const Vec4<R> o = {v[0], v[1], v[2], R(0)};
    if (D.collision_type == CONTACT_MIXED && D.keep_vmix) D.vmix[cell] = o;             // :403 (see grid_v_mixed_at)
    D.vout[cell] = o;                                                                   // :404 / :297
}

template <class R> __device__ __forceinline__ void gather_vec(const DevSim<R>& D, const Vec4<R>* field, const Stencil<R>& st,
                                                              const Nodes& nd, R* out) {
    out[0] = out[1] = out[2] = R(0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const R w = st.w[i][0] * st.w[j][1] * st.w[k][2];
                const Vec4<R> g = gld(field, nd.cell(i, j, k));
                out[0] += w * g.x; out[1] += w * g.y; out[2] += w * g.z;
            }
}

// grid_op_mixed2 + mixed3 + mixed4, one group of 32 lanes per particle of the hit list (the layout of k_contact_grad):
// lane n < 27 owns stencil node n (gather of v_mixed, scatter of the correction), every lane evaluates the particle's
// collide_mixed chain, 8 hits per workgroup share an LDS tile over the block of the first one.  (A per-chunk
// form walking a candidate-chunk list and a per-particle mask took 24 instead of 17 us: three more dependent loads and
// half as many busy workgroups.)
// CLOTH: the hit list names the particles with a contact face (k_cloth_hit_list) and mixed3 is the cloth primitive's collide_mixed
// (soft_cloth/engine/primitive/primitive_cloth.py:233-280) in physical units, its force splat onto the face's three vertices
template <class R, bool CLOTH>
__global__ __launch_bounds__(BLOCK) void k_contact_hits(DevSim<R> D, int f) {
    typedef typename pos_of<R>::type PX;
    __shared__ tile_t ctile[3 * TILE_WORDS];
    __shared__ double ext_acc[MAX_PRIMS * 6];
    __shared__ R fl_val[BLOCK * 4];                     // the flush's staging (flush_nodes_by_lanes)
    __shared__ unsigned fl_cell[BLOCK];
    // (tail reduction: no k_grid_op launch - forward_kinematics to frame f + 1 and the emptying of the next substep's hit counter ride here instead)
    if (D.fk_ride > 0 && blockIdx.x == gridDim.x - 1) {
        if ((int)threadIdx.x < D.fk_ride) prim_fk_step(D.prim_state + threadIdx.x * D.fk_stride, f, D.dt64);
        return;
    }
    SMAC_WAVE_T0();
    // one group of 32 lanes per hit; SMAC_HITS_PER_WAVE 1: the group is the lower half of a wave whose upper half idles (mask 0) - a wave then runs the chains of ONE hit
    constexpr int HPW = SMAC_HITS_PER_WAVE, WG_HITS = (BLOCK / 64) * HPW;
    const int grp = HPW == 2 ? (int)(threadIdx.x >> 5) : (int)(threadIdx.x >> 6), d = threadIdx.x & 31;
    const bool live = HPW == 2 || (threadIdx.x & 32) == 0;
    const double life = 1.0 / (double)(D.substeps - (f - D.frame_shift) % D.substeps);     // :425
    const int nwg = (int)gridDim.x - (D.fk_ride > 0 ? 1 : 0);            // (the last workgroup of a launch that carries forward_kinematics walks no hits)
    // This kernel is a chain of dependent round trips on an idle chip (a few thousand hits): the hit count, the first hit's block and this group's own hit record
    // are asked for TOGETHER (the records speculatively - the list holds D.Npad entries - and dropped when the count says there is none), not one after the other
    const int base0 = (int)blockIdx.x * WG_HITS;
    Hit h_next = D.hits[min(base0 + grp, D.Npad - 1)];
    int block_next = D.hits[min(base0, D.Npad - 1)].block;
    const int nh = *D.nhits;
    if (threadIdx.x < MAX_PRIMS * 6) ext_acc[threadIdx.x] = 0.0;
    __syncthreads();                                                                        // (waits for the three loads as well: the walk needs them at once anyway)
    if (D.zero_next_hits && blockIdx.x == 0 && threadIdx.x == 0) *D.nhits_next = 0;
    for (int base = base0; base < nh; base += nwg * WG_HITS) {
        for (int i = threadIdx.x; i < 3 * TILE_WORDS; i += BLOCK) ctile[i] = 0.0;           // (the barrier that publishes the zeros stands behind the chain: by then no load is in flight)
        const int wg_block = block_next;
        const int hi = base + grp;
        Hit h = {0, 0, 0, 0};
        if (hi < nh && live) h = h_next;
        {
            const int nb2 = base + nwg * WG_HITS;                                           // (a workgroup with a second round: the same, one round ahead)
            if (nb2 < nh) { h_next = D.hits[min(nb2 + grp, D.Npad - 1)]; block_next = D.hits[min(nb2, D.Npad - 1)].block; }
        }
        const int mask = h.mask, p = h.p;
        PX x[3] = {pos_mid<R>(), pos_mid<R>(), pos_mid<R>()};
        if (mask) load_pos(frame(D.S, f, D.Npad), D.Npad, p, x);
        Stencil<R> st;
        Nodes nd;
        stencil_at(D, x, st, nd, h.block);
        const int n = d < 27 ? d : 0;
        const int ni = n / 9, nj = (n / 3) % 3, nk = n % 3;
        const R wn = d < 27 ? (ni == 0 ? st.w[0][0] : (ni == 1 ? st.w[1][0] : st.w[2][0])) * (nj == 0 ? st.w[0][1] : (nj == 1 ? st.w[1][1] : st.w[2][1])) *
                                  (nk == 0 ? st.w[0][2] : (nk == 1 ? st.w[1][2] : st.w[2][2]))
                            : R(0);
        const unsigned cell = (unsigned)((ni == 0 ? nd.cx[0] : (ni == 1 ? nd.cx[1] : nd.cx[2])) + (nj == 0 ? nd.cy[0] : (nj == 1 ? nd.cy[1] : nd.cy[2])) +
                                         (nk == 0 ? nd.cz[0] : (nk == 1 ? nd.cz[1] : nd.cz[2])));
        Vec4<R> vm = {R(0), R(0), R(0), R(0)};
        bool has = false;
        if (mask && d < 27) {
            const Vec4<R> in_n = gld(D.vin, cell);
            vm = grid_v_mixed_at(D, in_n, cell);
            has = in_n.x > D.m_eps;
        }
        R v_tmp[3] = {wn * vm.x, wn * vm.y, wn * vm.z};                                     // mixed2
#pragma unroll
        for (int o = 16; o > 0; o >>= 1)
#pragma unroll
            for (int c = 0; c < 3; ++c) v_tmp[c] += __shfl_xor(v_tmp[c], o, 64);
        // mixed3: the push-out divides a signed distance by dt - f64 mode runs the whole chain in double; float32 mode keeps the DISTANCE in double and
        // everything else in float (collide_mixed_hybrid, smac_math.hpp: SMAC_CONTACT_HYBRID=0 builds the all-f64 chain of rounds 1-4 for A/B)
        constexpr bool HYB = SMAC_CONTACT_HYBRID && sizeof(R) == 4;
        const double x64[3] = {pos_get(x[0]), pos_get(x[1]), pos_get(x[2])};
        double v_tgt[3] = {(double)v_tmp[0], (double)v_tmp[1], (double)v_tmp[2]};
        float v32[3] = {(float)v_tmp[0], (float)v_tmp[1], (float)v_tmp[2]};
        // CLOTH instantiation: Hit::mask = sheet bits (0: has a contact face, 1: penetrated) | SDF-primitive band bits << 8.  A scene with SDF primitives
        // AND the sheet (round 4; BASELINE config C5 "mixed soft-rigid-cloth") walks the primitives in index order, then the sheet - the reference has no
        // simulator with both (softmac :421-429 loops the primitives, soft_cloth :419-428 has the one sheet): the composition is this build's.
        const int pmask = CLOTH ? (mask >> 8) & 15 : mask;
#pragma unroll 1
        for (int i = 0; i < D.P; ++i) {                                                     // mixed3: the SDF primitives in index order (:427)
            if (!((pmask >> i) & 1)) continue;
            const double* ps = D.prim_state + ((size_t)i * D.max_frames + f) * 13;
            double s13[13], ext[6] = {0, 0, 0, 0, 0, 0};
            for (int c = 0; c < 13; ++c) s13[c] = ps[c];
            if (HYB) {
                float e32[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                collide_mixed_hybrid<double, float>(D.prim64[i], *(const PrimTable<float>*)&D.prim[i], s13, x64, v32, (float)D.p_mass, D.dt64, life, e32);
                for (int c = 0; c < 6; ++c) ext[c] = (double)e32[c];
                for (int c = 0; c < 3; ++c) v_tgt[c] = (double)v32[c];
            } else
                collide_mixed(D.prim64[i], s13, x64, v_tgt, (double)D.p_mass, D.dt64, life, ext);
            if (d < 6) {                                                                    // lane c adds component c
                const double e = d == 0 ? ext[0] : (d == 1 ? ext[1] : (d == 2 ? ext[2] : (d == 3 ? ext[3] : (d == 4 ? ext[4] : ext[5]))));
                if (e != 0.0) __hip_atomic_fetch_add(ext_acc + i * 6 + d, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (CLOTH && (mask & 1)) {                                                          // mixed3, soft_cloth :419-428: the sheet, on the velocity the primitives left
            const ClothDev& Cl = D.cloth;
            const double sc = Cl.par.scale;
            const int* vid = Cl.faces + 3 * h.pad;
            double xv[3][3], vv[3][3];
            for (int i = 0; i < 3; ++i)
                for (int c = 0; c < 3; ++c) {
                    xv[i][c] = Cl.pos[((size_t)f * Cl.V + vid[i]) * 3 + c];
                    vv[i][c] = Cl.vel[((size_t)f * Cl.V + vid[i]) * 3 + c];
                }
            const double pp[3] = {sc * x64[0], sc * x64[1], sc * x64[2]};
            double vio[3] = {sc * v_tgt[0], sc * v_tgt[1], sc * v_tgt[2]}, cf[3], wb[3];
            if (cloth_collide_mixed<double>(Cl.par, xv, vv, pp, vio, (double)D.p_mass * sc * sc, D.dt64, life, (mask >> 1) & 1, cf, wb)) {
                for (int c = 0; c < 3; ++c) v_tgt[c] = vio[c] / sc;
                if (d < 9) {                                                                // :276-278, lane d adds component d % 3 of vertex d / 3
                    const int vi = d / 3, c = d % 3;
                    const double wv = vi == 0 ? wb[0] : (vi == 1 ? wb[1] : wb[2]), cc = c == 0 ? cf[0] : (c == 1 ? cf[1] : cf[2]);
                    atomic_add(Cl.ext_f + (size_t)(vi == 0 ? vid[0] : (vi == 1 ? vid[1] : vid[2])) * 3 + c, cc * wv);
                }
            }
        }
        __syncthreads();                                                                    // tile zeroed everywhere
        R late[3] = {R(0), R(0), R(0)};                                                     // a node outside the tile: its global atomics leave AFTER the barrier below
        bool is_late = false;                                                               // (adds queued in the memory pipeline hold up their wave's next memory instruction, and with the wave everybody at a barrier)
        if (mask && d < 27 && has) {                                                        // mixed4, alpha = 2 (:437)
            const bool in_tile = h.block == wg_block && ((nd.okx >> ni) & (nd.oky >> nj) & (nd.okz >> nk) & 1) != 0;
            const int tw = (ni == 0 ? nd.tx[0] : (ni == 1 ? nd.tx[1] : nd.tx[2])) + (nj == 0 ? nd.ty[0] : (nj == 1 ? nd.ty[1] : nd.ty[2])) +
                           (nk == 0 ? nd.tz[0] : (nk == 1 ? nd.tz[1] : nd.tz[2]));
            is_late = !in_tile;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const R val = -R(2) * wn * (R)((double)v_tmp[c] - v_tgt[c]);
                if (in_tile) lds_add(ctile + tw + c * TILE_WORDS, val);
                else late[c] = val;
            }
        }
        __syncthreads();
        if (__syncthreads_or(is_late)) {                                                    // (hits of another block than the tile's: by lanes as the tile's)
            const Vec4<R> lo = {late[0], late[1], late[2], R(0)};
            flush_nodes_by_lanes(D.vout, lo, cell, fl_val, fl_cell);
        }
        {
            static_assert(TILE_WORDS <= BLOCK, "one tile node per thread in the flush");
            Vec4<R> o = {R(0), R(0), R(0), R(0)};
            unsigned c2 = 0u;
            if (threadIdx.x < TILE_WORDS) {
                const int nb = D.nb, idx = threadIdx.x;
                const int bz = wg_block % nb, by = (wg_block / nb) % nb, bx = wg_block / (nb * nb);
                o = Vec4<R>{(R)ctile[idx], (R)ctile[TILE_WORDS + idx], (R)ctile[2 * TILE_WORDS + idx], R(0)};
                if (o.x != R(0) || o.y != R(0) || o.z != R(0)) c2 = (unsigned)cell_of(nb, 4 * bx + idx / TSX, 4 * by + (idx / TSY) % TW, 4 * bz + idx % TW);
            }
            flush_nodes_by_lanes(D.vout, o, c2, fl_val, fl_cell);                             // (its closing barrier: the tile may be zeroed again)
        }
    }
    __syncthreads();
    if (threadIdx.x < D.P * 6 && ext_acc[threadIdx.x] != 0.0) atomic_add(D.ext_f + threadIdx.x, ext_acc[threadIdx.x]);
    SMAC_WAVE_HIST(0);
}

// one particle of g2p :299-318: gather from the staged tile `gt` (global fallback for a lane that drifted out of it), write frame f+1
template <class R>
__device__ __forceinline__ void g2p_particle(const DevSim<R>& D, const Chunk& ch, int p, const typename pos_of<R>::type* x, const Vec4<R>* gt, R* Sn,
                                             typename pos_of<R>::type* xn_o = nullptr, R* nv_o = nullptr, R* nC_o = nullptr) {   // (k_g2p_p2g: x, v, C of frame f + 1 in registers too)
    Stencil<R> st;
    Nodes nd;
    stencil_at_p(D, x, st, nd, ch.block);
    const bool all_in = (nd.okx & nd.oky & nd.okz) == 7;
    if (!all_in) {          // stencil left the chunk's tile: how far has this particle drifted?
        const int nb = D.nb;
        const int pb[3] = {st.base[0] >> 2, st.base[1] >> 2, st.base[2] >> 2};
        const int cbk[3] = {ch.block / (nb * nb), (ch.block / nb) % nb, ch.block % nb};
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (pb[d] < cbk[d] - 1 || pb[d] > cbk[d] + 1) *D.drift_flag = 1;            // beyond the active halo
    }
    // Separable evaluation of the stencil sums: M0 = sum w g and the first moments Mx,My,Mz = sum w {i,j,k} g are
    // built along z, then y, then x (240 FMAs instead of 459 for the flat 27-node form); then
    // new_v = M0, new_C[c][d] = 4 inv_dx (M_d[c] - f_d M0[c])   (dpos = offset - fx, mpm_simulator.py:308-314).
    R M0[3] = {R(0), R(0), R(0)}, Mx[3] = {R(0), R(0), R(0)}, My[3] = {R(0), R(0), R(0)}, Mz[3] = {R(0), R(0), R(0)};
    const R wz1 = st.w[1][2], wz2 = R(2) * st.w[2][2], wy1 = st.w[1][1], wy2 = R(2) * st.w[2][1];
    // Two copies of the gather, chosen per WAVE: without a drifted lane every node is a plain ds_read_b128 of the tile.  Written as ONE
    // loop with a per-lane choice, the compiler folds "LDS value, overridden from global memory for drifted lanes" into a flat load through
    // a selected pointer - 27 flat_load per particle also for the waves that never leave LDS (found in the ISA, round 2).
    auto gather = [&](auto mixed_tag) {
        constexpr bool MIXED = decltype(mixed_tag)::value;
#if SMAC_G2P_ROLLED
        // x-planes as a REAL loop (9 records live instead of 27), the plane's x weight / offsets rotating through registers as in p2g_grad_particle
        R wxi = st.w[0][0], wx1 = st.w[1][0], wx2 = st.w[2][0];
        int cxi = nd.cx[0], cx1 = nd.cx[1], cx2 = nd.cx[2];
        int txi = nd.tx[0], tx1 = nd.tx[1], tx2 = nd.tx[2];
#pragma unroll 1
        for (int i = 0; i < 3; ++i) {
            R s0[3] = {R(0), R(0), R(0)}, sy[3] = {R(0), R(0), R(0)}, sz[3] = {R(0), R(0), R(0)};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                Vec4<R> g0, g1, g2;
                if (MIXED && !all_in) {
                    g0 = gld(D.vout, (unsigned)(cxi + nd.cy[j] + nd.cz[0])); g1 = gld(D.vout, (unsigned)(cxi + nd.cy[j] + nd.cz[1])); g2 = gld(D.vout, (unsigned)(cxi + nd.cy[j] + nd.cz[2]));
                } else { g0 = gt[txi + nd.ty[j] + nd.tz[0]]; g1 = gt[txi + nd.ty[j] + nd.tz[1]]; g2 = gt[txi + nd.ty[j] + nd.tz[2]]; }
                const R r0[3] = {st.w[0][2] * g0.x + st.w[1][2] * g1.x + st.w[2][2] * g2.x, st.w[0][2] * g0.y + st.w[1][2] * g1.y + st.w[2][2] * g2.y,
                                 st.w[0][2] * g0.z + st.w[1][2] * g1.z + st.w[2][2] * g2.z};
                const R r1[3] = {wz1 * g1.x + wz2 * g2.x, wz1 * g1.y + wz2 * g2.y, wz1 * g1.z + wz2 * g2.z};
                const R wyj = st.w[j][1];
#pragma unroll
                for (int c = 0; c < 3; ++c) { s0[c] += wyj * r0[c]; sz[c] += wyj * r1[c]; }
                if (j == 1) { for (int c = 0; c < 3; ++c) sy[c] += wy1 * r0[c]; }
                if (j == 2) { for (int c = 0; c < 3; ++c) sy[c] += wy2 * r0[c]; }
            }
            const R fi = R(i);
#pragma unroll
            for (int c = 0; c < 3; ++c) { M0[c] += wxi * s0[c]; My[c] += wxi * sy[c]; Mz[c] += wxi * sz[c]; Mx[c] += fi * wxi * s0[c]; }
            wxi = wx1; wx1 = wx2; cxi = cx1; cx1 = cx2; txi = tx1; tx1 = tx2;
        }
        return;
#endif
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            R s0[3] = {R(0), R(0), R(0)}, sy[3] = {R(0), R(0), R(0)}, sz[3] = {R(0), R(0), R(0)};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                Vec4<R> g0, g1, g2;
                if (MIXED && !all_in) { g0 = gld(D.vout, nd.cell(i, j, 0)); g1 = gld(D.vout, nd.cell(i, j, 1)); g2 = gld(D.vout, nd.cell(i, j, 2)); }
                else { g0 = gt[nd.tile(i, j, 0)]; g1 = gt[nd.tile(i, j, 1)]; g2 = gt[nd.tile(i, j, 2)]; }
                const R r0[3] = {st.w[0][2] * g0.x + st.w[1][2] * g1.x + st.w[2][2] * g2.x, st.w[0][2] * g0.y + st.w[1][2] * g1.y + st.w[2][2] * g2.y,
                                 st.w[0][2] * g0.z + st.w[1][2] * g1.z + st.w[2][2] * g2.z};
                const R r1[3] = {wz1 * g1.x + wz2 * g2.x, wz1 * g1.y + wz2 * g2.y, wz1 * g1.z + wz2 * g2.z};
                const R wyj = st.w[j][1];
#pragma unroll
                for (int c = 0; c < 3; ++c) { s0[c] += wyj * r0[c]; sz[c] += wyj * r1[c]; }
                if (j == 1) { for (int c = 0; c < 3; ++c) sy[c] += wy1 * r0[c]; }
                if (j == 2) { for (int c = 0; c < 3; ++c) sy[c] += wy2 * r0[c]; }
            }
            const R wxi = st.w[i][0];
#pragma unroll
            for (int c = 0; c < 3; ++c) { M0[c] += wxi * s0[c]; My[c] += wxi * sy[c]; Mz[c] += wxi * sz[c]; }
            if (i == 1) { for (int c = 0; c < 3; ++c) Mx[c] += wxi * s0[c]; }
            if (i == 2) { for (int c = 0; c < 3; ++c) Mx[c] += R(2) * wxi * s0[c]; }
        }
    };
    if (__all(all_in)) gather(std::false_type{});
    else gather(std::true_type{});
    R nv[3], nC[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        nv[c] = M0[c];
        nC[3 * c + 0] = Mx[c] - st.fx[0] * M0[c];
        nC[3 * c + 1] = My[c] - st.fx[1] * M0[c];
        nC[3 * c + 2] = Mz[c] - st.fx[2] * M0[c];
    }
    const R four_inv_dx = R(4) * D.inv_dx;
    bool leaves = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Sn[rowoff(CV + c, p, D.Npad)] = nv[c];
        const typename pos_of<R>::type xn = pos_advance(x[c], D.dt64, nv[c]);          // :318
        store_pos(Sn, D.Npad, p, c, xn);
        if (xn_o) { xn_o[c] = xn; nv_o[c] = nv[c]; }
        // the position just written is scattered by the NEXT substep's P2G: if that substep keeps this binning, the new
        // base must still lie in the blocks around the chunk's own (the only ones that are cleared, reduced and swept)
        int nbase = pos_base(xn, D.n);
        if (c == 0 && D.slab_base_lo <= D.slab_base_hi && (nbase < D.slab_base_lo || nbase > D.slab_base_hi)) D.drift_flag[2] = 1;   // left the slab's shared planes
        nbase = nbase < 0 ? 0 : (nbase > D.n - 3 ? D.n - 3 : nbase);
        const int cbk = c == 0 ? ch.block / (D.nb * D.nb) : (c == 1 ? (ch.block / D.nb) % D.nb : ch.block % D.nb);
        leaves |= (nbase >> 2) < cbk - 1 || (nbase >> 2) > cbk + 1;
        // tail reduction: a block's arrival count covers the chunks of the 27 blocks around it, so the WHOLE stencil (base .. base + 2) of the next P2G must stay
        // inside the blocks cbk - 1 .. cbk + 1 (without it a base in block cbk + 1 may reach cbk + 2, which k_grid_op swept and no count covers)
        leaves |= SMAC_TAIL_BUILD && D.tail_rule && ((nbase + 2) >> 2) > cbk + 1;
    }
    if (leaves && D.check_next) *D.drift_flag = 1;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const R cc = four_inv_dx * nC[c];
        Sn[rowoff(CC + c, p, D.Npad)] = cc;
        if (nC_o) nC_o[c] = cc;
    }
}

// forward_kinematics :280-283 of primitive `i` from frame f to f + 1, and its adjoint (13 inputs -> 7 outputs, forward-mode duals: thread `dir` owns input `dir`)
__device__ __forceinline__ void prim_fk_step(double* state, int f, double dt) {
    double* s = state + (size_t)f * 13;
    double o[7];
    forward_kinematics(s, dt, o);
    for (int i = 0; i < 7; ++i) s[13 + i] = o[i];
}
__device__ __forceinline__ void prim_fk_step_grad(const double* state, double* grad, int f, double dt, int dir) {
    const double* s = state + (size_t)f * 13;
    Dual<double> sd[13], o[7];
    for (int i = 0; i < 13; ++i) sd[i] = Dual<double>(s[i], i == dir ? 1.0 : 0.0);
    forward_kinematics(sd, dt, o);
    double acc = 0.0;
    for (int i = 0; i < 7; ++i) acc += grad[(size_t)(f + 1) * 13 + i] * o[i].d;
    grad[(size_t)f * 13 + dir] += acc;
}

// SAVE: the grid checkpoint of the frame (k_grid_save's work: 20 s G_t bytes, a 9 us launch of its own that only waits for memory) is done by the
// first D.save_blocks workgroups of THIS launch - it reads the same finished grid and touches nothing k_g2p reads ({m,p} is zeroed, v_out only
// copied), so the two run side by side and a kernel boundary goes.  The hit counter cannot be emptied here while the save part still reads it:
// the counters of even and odd frames alternate, and this launch empties the NEXT frame's.
template <class R, bool SAVE>
__global__ __launch_bounds__(BLOCK, (sizeof(R) == 4 ? SMAC_OCC_G2P : 2)) void k_g2p(DevSim<R> D, int f) {
    __shared__ Vec4<R> gt[PTILE];
    int bid = (int)blockIdx.x;
    if (D.fk_ride > 0 && blockIdx.x == gridDim.x - 1) {       // the velocity-controlled primitives step to frame f + 1 here (nothing in substep f reads that frame)
        if ((int)threadIdx.x < D.fk_ride) prim_fk_step(D.prim_state + threadIdx.x * D.fk_stride, f, D.dt64);
        return;
    }
    if (SAVE) {
        if (bid == 0 && threadIdx.x == 0) {
            D.last_counts[0] = *D.nhits; D.last_counts[1] = *D.ncand;
            *D.nhits_next = 0; *D.ncand = 0;
        }
        if (bid < D.save_blocks) {
            grid_save_block(D, D.save_blocks, D.save_ck, D.save_hits, D.save_nhits, D.save_hit_cap);
            return;
        }
        bid -= D.save_blocks;
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {      // last kernel of the substep: hand the contact lists back empty
        D.last_counts[0] = *D.nhits; D.last_counts[1] = *D.ncand;
        *D.nhits = 0; *D.nhits_next = 0; *D.ncand = 0;
    }
    const int cid = xcd_chunk_at(bid, D.nchunks);
    if (cid >= D.nchunks) return;
    SMAC_CHUNK_PROLOGUE_AT(cid)
    const R* Sf = frame(D.S, f, D.Npad);
    R* Sn = frame(D.S, f + 1, D.Npad);
    typename pos_of<R>::type x[3];
    SMAC_PHASE(24, valid);
    load_pos(Sf, D.Npad, p, x);                     // issued before the tile load's barrier: one round trip, not two
    gather_tile_load_p(D, D.vout, ch.block, gt);
    __syncthreads();
    SMAC_PHASE(25, valid);
    if (!valid) return;
    g2p_particle(D, ch, p, x, gt, Sn);
    SMAC_PHASE(26, true);
}

// G2P of substep f and P2G of substep f + 1 in one launch (round 4; the forward counterpart of k_p2g_g2p_grad).  Between two re-sorts a particle keeps its
// chunk, and x, v, C of frame f + 1 that G2P has just produced are exactly what P2G of the next substep loads first: they stay in registers (15 rows of
// reads, a kernel boundary and a workgroup prologue less per forward substep; the rows are still written - every frame stays readable).  P2G of
// substep f + 1 scatters into the slabs and, for what leaves a tile, into D.vdrift - neither is read by G2P or by the checkpoint save of substep f in the
// first workgroups of this launch; its hits go to the OTHER of two hit lists (D.hits_next / D.nhits_next: the save part copies this substep's).  The host
// has run forward_kinematics to frame f + 1 and emptied the next counter inside k_grid_op's launch of substep f.  SAVE: as k_g2p<R, true>.
template <class R, bool SAVE>
__global__ __launch_bounds__(BLOCK, (sizeof(R) == 4 ? SMAC_OCC_G2P : 2)) void k_g2p_p2g(DevSim<R> D, int f) {
    typedef typename ScatterTile<R>::word W;
    typedef typename pos_of<R>::type PX;
    __shared__ __attribute__((aligned(16))) double tile_raw[4 * PTILE];
    __shared__ Vec4<R> gt[PTILE];
    __shared__ R smax[4];
    __shared__ R ps_wg[MAX_PRIMS * 13];
    int bid = (int)blockIdx.x;
    if (SAVE) {
        if (bid == 0 && threadIdx.x == 0) { D.last_counts[0] = *D.nhits; D.last_counts[1] = *D.ncand; *D.ncand = 0; }
        if (bid < D.save_blocks) {
            grid_save_block(D, D.save_blocks, D.save_ck, D.save_hits, D.save_nhits, D.save_hit_cap);
            return;
        }
        bid -= D.save_blocks;
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        D.last_counts[0] = *D.nhits; D.last_counts[1] = *D.ncand;
        *D.nhits = 0; *D.ncand = 0;
    }
    const int cid = xcd_chunk_at(bid, D.nchunks);
    if (cid >= D.nchunks) return;
    if (D.any_contact && threadIdx.x < D.P * 13)      // primitive states of frame f + 1 for the band test (published by the barrier below)
        ps_wg[threadIdx.x] = (R)D.prim_state[((size_t)(threadIdx.x / 13) * D.max_frames + f + 1) * 13 + threadIdx.x % 13];
    SMAC_CHUNK_PROLOGUE_AT(cid)
    lds_zero16(tile_raw, 4 * PTILE * (int)((sizeof(R) == 4 && ch.count <= SPARSE_MAX) ? sizeof(double) : sizeof(W)));
    const R* Sf = frame(D.S, f, D.Npad);
    R* Sn = frame(D.S, f + 1, D.Npad);
    PX x[3];
    load_pos(Sf, D.Npad, p, x);
    gather_tile_load_p(D, D.vout, ch.block, gt);
    __syncthreads();
    PX xn[3] = {pos_mid<R>(), pos_mid<R>(), pos_mid<R>()};
    R nv[3] = {R(0), R(0), R(0)}, nC[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) nC[c] = R(0);
    if (valid) g2p_particle(D, ch, p, x, gt, Sn, xn, nv, nC);
    p2g_body<R, true, false, false, true>(D, f + 1, ch, cid, t, valid, p, tile_raw, smax, ps_wg, xn, nv, nC, D.hits_next, D.nhits_next);
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
// adjoint of the three per-dimension weight factors -> adjoint of fx
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
        // the weight derivatives are recomputed from fx here (mpm_simulator.py:217) instead of being kept live across the gather
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const R fx = st.fx[d];
            gfx[d] += g[0][d] * (fx - R(1.5)) + g[1][d] * (R(-2) * (fx - R(1))) + g[2][d] * (fx - R(0.5));
        }
    }
};

// G2P adjoint of one chunk, from the adjoint of the substep's outputs held in registers: gx1 = x'.grad, gnv = v'.grad + dt x'.grad, gC1 = 4 inv_dx C'.grad
// (all zero for an idle lane).  `x` are the positions the substep started from, `gt` the staged grid_v_out tile, `tile_raw` the zeroed scatter
// tile; writes x.grad to `Af`, the node adjoints to the chunk's slab.  Every thread of the workgroup calls it (two barriers).
template <class R, bool ACC_X>
__device__ __forceinline__ void g2p_grad_chunk(const DevSim<R>& D, const Chunk& ch, int p, int t, bool valid, const typename pos_of<R>::type* x,
                                               const R* gx1, const R* gnv, const R* gC1, R* Af, double* tile_raw, const Vec4<R>* gt, R* smax) {
    typedef typename ScatterTile<R>::word W;
    W* const tile = (W*)tile_raw;
    double* const tile64 = tile_raw;
    const bool sparse = sizeof(R) == 4 && ch.count <= SPARSE_MAX;       // f64 words instead of fixed point (see SPARSE_MAX)
    (void)t;
    R bound = R(0);                       // no single scattered component of this particle exceeds it
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c)      // |w (gnv + gC (offset - fx))|
            bound = maxc(bound, R(W_MAX) * abs_(gnv[c]) + R(WD_MAX) * (abs_(gC1[3 * c]) + abs_(gC1[3 * c + 1]) + abs_(gC1[3 * c + 2])));
    }
    R to_tile, from_tile;
    const bool fused_ = Af == D.Af_prev;
    (void)fused_;
    tile_scale<R>(bound, smax, to_tile, from_tile);                                        // one barrier (all threads)
    SMAC_PHASE(6, valid && fused_);
    if (valid) {
        Stencil<R> st;
        Nodes nd;
        stencil_at_p(D, x, st, nd, ch.block);
        WGrad<R> wg;
        wg.zero();
        R gfx[3] = {R(0), R(0), R(0)};
        {
            // Everything is affine in the node offset:
            //   scatter  t(i,j,k)[c] = T0[c] + i gC[c][0] + j gC[c][1] + k gC[c][2],   T0 = gnv - gC f       (3 adds per node)
            //   weight adjoint  Q(i,j,k) = g(i,j,k) . t(i,j,k), folded along z, y, x into Gx,Gy,Gz
            //   dpos adjoint    gfx[d] = - sum_c gC[c][d] M0[c],  M0 = sum w g   (one 3-vector, not 27 x 9 products)
            R T0[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) T0[c] = gnv[c] - (gC1[3 * c] * st.fx[0] + gC1[3 * c + 1] * st.fx[1] + gC1[3 * c + 2] * st.fx[2]);
            R M0[3] = {R(0), R(0), R(0)};
            R gwx[3] = {R(0), R(0), R(0)};
            const int tbase = nd.tx[0] + nd.ty[0] + nd.tz[0];
            // One x-plane (9 nodes) per trip: rolled, so only 9 gathered records are live at a time.  MIXED = some lane
            // of the wave drifted out of its block since the last sort: same code with a per-node choice between the LDS
            // tiles and global memory - one pass for all lanes instead of a fast and a slow path run one after the other.
            auto planes = [&](auto mixed_tag, auto* tl, const R to_tl) {
                constexpr bool MIXED = decltype(mixed_tag)::value;
#pragma unroll 1
                for (int i = 0; i < 3; ++i) {
                    const R fi = (R)i;
                    const R wxi = i == 0 ? st.w[0][0] : (i == 1 ? st.w[1][0] : st.w[2][0]);
                    const int ti0 = tbase + i * PSX;
                    const int cxi = i == 0 ? nd.cx[0] : (i == 1 ? nd.cx[1] : nd.cx[2]);
                    const int okxi = (nd.okx >> i) & 1;
                    R ti[3] = {T0[0] + fi * gC1[0], T0[1] + fi * gC1[3], T0[2] + fi * gC1[6]};
                    R s0[3] = {R(0), R(0), R(0)};
                    R gxi = R(0);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const R wij = wxi * st.w[j][1];
                        R tj[3] = {ti[0] + R(j) * gC1[1], ti[1] + R(j) * gC1[4], ti[2] + R(j) * gC1[7]};
                        R aq = R(0);                    // sum_k Q wz_k
                        R r0[3] = {R(0), R(0), R(0)};   // sum_k wz_k g
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const int tw = ti0 + j * PSY + k;
                            const bool in = !MIXED || (okxi & (nd.oky >> j) & (nd.okz >> k) & 1) != 0;
                            const unsigned cell = (unsigned)(cxi + nd.cy[j] + nd.cz[k]);
                            Vec4<R> g = gt[in ? tw : 0];                 // unconditional LDS read (see k_g2p)
                            if (MIXED && !in) g = gld(D.vout, cell);
                            const R tk[3] = {tj[0] + R(k) * gC1[2], tj[1] + R(k) * gC1[5], tj[2] + R(k) * gC1[8]};
                            const R w = wij * st.w[k][2];
                            if (in) {
#pragma unroll
                                for (int c = 0; c < 3; ++c) tile_add(tl + tw + c * PTILE, (w * to_tl) * tk[c]);
                            } else {
#pragma unroll
                                for (int c = 0; c < 3; ++c) gatomic(D.aout, cell, c, w * tk[c]);
                            }
                            const R Q = g.x * tk[0] + g.y * tk[1] + g.z * tk[2];
                            aq += Q * st.w[k][2];
                            wg.g[k][2] += Q * wij;
                            r0[0] += st.w[k][2] * g.x; r0[1] += st.w[k][2] * g.y; r0[2] += st.w[k][2] * g.z;
                        }
                        gxi += aq * st.w[j][1];
                        wg.g[j][1] += aq * wxi;
#pragma unroll
                        for (int c = 0; c < 3; ++c) s0[c] += st.w[j][1] * r0[c];
                    }
                    gwx[0] += i == 0 ? gxi : R(0); gwx[1] += i == 1 ? gxi : R(0); gwx[2] += i == 2 ? gxi : R(0);
#pragma unroll
                    for (int c = 0; c < 3; ++c) M0[c] += wxi * s0[c];
                }
            };
            if (sparse) planes(std::true_type{}, tile64, R(1));
            else if (__all((nd.okx & nd.oky & nd.okz) == 7)) planes(std::false_type{}, tile, to_tile);
            else planes(std::true_type{}, tile, to_tile);
#pragma unroll
            for (int a = 0; a < 3; ++a) wg.g[a][0] += gwx[a];
#pragma unroll
            for (int d = 0; d < 3; ++d) gfx[d] -= gC1[d] * M0[0] + gC1[3 + d] * M0[1] + gC1[6 + d] * M0[2];   // dpos = offset - fx
        }
        wg.to_fx(st, gfx);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const R g = gx1[d] + D.inv_dx * gfx[d];
            if (ACC_X) Af[rowoff(CX + d, p, D.Npad)] += g;
            else Af[rowoff(CX + d, p, D.Npad)] = g;
        }
    }
    SMAC_PHASE(7, valid && fused_);            // 27-node gather + scatter of the G2P adjoint
    __syncthreads();
    SMAC_PHASE(8, valid && fused_);
    const bool tail = SMAC_TAIL_BUILD && D.tail_on != 0;               // (launch-uniform)
    if (sparse) { tile_store<R, 3>(D, tile64, R(1), R(1), -1, tail); tile_flush_shell<R, 3>(D, tile64, D.aout, ch.block, R(1), R(1)); }
    else { tile_store<R, 3>(D, tile, from_tile, from_tile, -1, tail); tile_flush_shell<R, 3>(D, tile, D.aout, ch.block, from_tile, from_tile); }
    SMAC_PHASE(9, valid && fused_);
    // tail reduction (see tail_arrive): the chunk that completes a block's count finishes grid_v_out.grad there and pushes it through grid_op's node adjoint
    if (tail) tail_arrive(D, ch.block, (int*)smax, [&](int B) { tail_block_bwd(D, B); });
}


template <class R, bool ACC_X>
__global__ __launch_bounds__(BLOCK, (sizeof(R) == 4 ? SMAC_OCC_G2PG : 2)) void k_g2p_grad(DevSim<R> D, int f) {
    typedef typename ScatterTile<R>::word W;
    __shared__ __attribute__((aligned(16))) double tile_raw[3 * PTILE];
    __shared__ Vec4<R> gt[PTILE];
    __shared__ R smax[4];
    SMAC_CHUNK_PROLOGUE
    W* const tile = (W*)tile_raw;
    double* const tile64 = tile_raw;
    const bool sparse = sizeof(R) == 4 && ch.count <= SPARSE_MAX;
    lds_zero16(tile_raw, 3 * PTILE * (int)(sparse ? sizeof(double) : sizeof(W)));
    gather_tile_load_p(D, D.vout, ch.block, gt);        // (its barrier is the one inside tile_scale: the particle loads
                                                    //  that follow are then in flight together with the tile's)
    typename pos_of<R>::type x[3] = {pos_mid<R>(), pos_mid<R>(), pos_mid<R>()};
    R gx1[3] = {R(0), R(0), R(0)}, gnv[3] = {R(0), R(0), R(0)}, gC1[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) gC1[c] = R(0);
    if (valid) {
        const R* Sf = frame(D.S, f, D.Npad);
        const R* An = D.An;
        const int pa = D.An_map ? D.An_map[p] : p;
        R gv1[3];
        load_pos(Sf, D.Npad, p, x);
        load_vec(An, CX, 3, D.Npad, pa, gx1);
        load_vec(An, CV, 3, D.Npad, pa, gv1);
        load_vec(An, CC, 9, D.Npad, pa, gC1);
        const R four_inv_dx = R(4) * D.inv_dx;
#pragma unroll
        for (int c = 0; c < 3; ++c) gnv[c] = gv1[c] + D.dt * gx1[c];                        // x' = x + dt v'
#pragma unroll
        for (int c = 0; c < 9; ++c) gC1[c] *= four_inv_dx;
    }
    g2p_grad_chunk<R, ACC_X>(D, ch, p, t, valid, x, gx1, gnv, gC1, D.Af, tile_raw, gt, smax);
}

// completes the G2P-adjoint scatter: grid_v_out.grad += sum of overlapping slabs
// HALO (round 5): the library's slab loop, backward exchange of grid_v_out.grad - this launch writes the shared planes' partial sums into the send buffer itself
// (k_halo_pack2's work) and k_grid_op_grad<.., HALO> adds what arrived (k_halo_unpack_add2's); only where no contact primitive can reach a shared plane: the
// contact adjoint in between would gather totals there
template <class R, bool HALO = false>
__global__ __launch_bounds__(BLOCK) void k_reduce_aout(DevSim<R> D) {
    int b, l, i, j, k;
    size_t cell;
    if (!active_cell(D, b, l, cell, i, j, k)) return;
    Vec4<R> acc = D.aout[cell];
    slab_reduce(D, b, l, acc);
    D.aout[cell] = acc;
    if (HALO && D.halo_hs.count) {
        const size_t total = (size_t)D.halo_np * D.n * D.n;
        for (int s = 0; s < D.halo_hs.count; ++s) {
            const int pi = i - D.halo_hs.plane0[s];
            if ((unsigned)pi < (unsigned)D.halo_np) D.halo_send[(size_t)D.halo_hs.slot[s] * total + ((size_t)pi * D.n + j) * D.n + k] = acc;
        }
    }
}

// k_reduce_aout + the adjoint of grid_op / grid_op_mixed1 in one pass over the active cells (whole-substep path without grid-node
// contact).  grid_op_mixed1's adjoint is linear in grid_v_mixed.grad, so it does not have to wait for the contact adjoint: this
// kernel sends grid_v_out.grad through it, and k_contact_grad<DIRECT> sends each of its own sparse grid_v_mixed.grad contributions
// through the same node map and adds them to grid_v_in.grad / grid_m.grad (one kernel and the dense grid_v_mixed.grad field less).
template <class R>
__global__ __launch_bounds__(BLOCK) void k_reduce_grid_grad(DevSim<R> D) {
    if (D.fk_ride > 0 && blockIdx.x >= gridDim.x - D.fk_ride) {                   // forward_kinematics.grad of this substep, one workgroup per primitive (:367-369)
        const int prim = (int)(blockIdx.x - (gridDim.x - D.fk_ride));
        if (threadIdx.x < 13) prim_fk_step_grad(D.prim_state + prim * D.fk_stride, D.prim_grad + prim * D.fk_stride, D.cur_frame, D.dt64, (int)threadIdx.x);
        return;
    }
    int b, l, i, j, k;
    size_t cell;
    if (!active_cell(D, b, l, cell, i, j, k)) return;
    // a block that held no mass in this frame (flag written with its checkpoint): no particle's stencil reached it, so nothing was scattered to it and
    // nothing will gather its grid_v_in.grad - no reads, no writes (what the array holds there is never looked at: gathers read touched nodes only)
    if (D.ck_flags && D.ck_flags[active_slot(D)]) return;
    Vec4<R> acc = D.aout[cell];
    const Vec4<R> in = D.vin[cell];
    slab_reduce(D, b, l, acc);
    if (D.any_contact) D.aout[cell] = acc;           // the contact adjoint gathers grid_v_out.grad at the nodes of its hits
    R g[3] = {acc.x, acc.y, acc.z};
    D.ain[cell] = grid_op_node_adjoint(D, in, i, j, k, g);
}

// The same, with the forward grid of the NEXT substep of the sweep (the one before this in time) restored from its checkpoint by the second half of the
// launch (k_grid_restore's work).  That grid goes to the OTHER of two buffer sets - this substep's reduction and contact adjoint still read the
// current one - and the fused backward step that follows gathers grid_v_out from there; then the sets change roles.  The hit list is not copied:
// the contact adjoint of that substep walks the filed list in place.  One launch and 9 us of waiting for memory less per backward substep.
template <class R> struct GridSet { Vec4<R> *vin, *vmix, *vout, *aout; };
// one wave restores the forward grid of active slot `a` of the NEXT frame of the sweep from its checkpoint into the buffer set `nx`
template <class R> __device__ __forceinline__ void restore_ahead_slot(const DevSim<R>& D, const GridSet<R>& nx, const Vec4<R>* ck, int a) {
    if (a >= D.nactive) return;
    const int l = threadIdx.x & 63;
    const size_t cell = (size_t)D.active[a] * 64 + l;
    const Vec4<R>* src = ck + (size_t)a * CK_WORDS + l;
    const Vec4<R> z = {R(0), R(0), R(0), R(0)};
    if (D.ck_flags_next && D.ck_flags_next[a]) { nx.vin[cell] = z; nx.vout[cell] = z; }
    else { nx.vin[cell] = src[0]; nx.vout[cell] = src[64]; }
    nx.aout[cell] = z;                               // g2p.grad's drifted lanes add to it
}
template <class R>
__global__ __launch_bounds__(BLOCK) void k_reduce_grid_grad_ahead(DevSim<R> D, GridSet<R> nx, const Vec4<R>* ck) {
    if (D.fk_ride > 0 && blockIdx.x >= gridDim.x - D.fk_ride) {
        const int prim = (int)(blockIdx.x - (gridDim.x - D.fk_ride));
        if (threadIdx.x < 13) prim_fk_step_grad(D.prim_state + prim * D.fk_stride, D.prim_grad + prim * D.fk_stride, D.cur_frame, D.dt64, (int)threadIdx.x);
        return;
    }
    const int half = (int)((gridDim.x - (D.fk_ride > 0 ? D.fk_ride : 0)) >> 1);
    if ((int)blockIdx.x < half) {
        int b, l, i, j, k;
        size_t cell;
        if (!active_cell(D, b, l, cell, i, j, k)) return;
        if (D.ck_flags && D.ck_flags[active_slot(D)]) return;            // (no mass in this frame: see k_reduce_grid_grad)
        Vec4<R> acc = D.aout[cell];
        const Vec4<R> in = D.vin[cell];
        slab_reduce(D, b, l, acc);
        if (D.any_contact) D.aout[cell] = acc;
        R g[3] = {acc.x, acc.y, acc.z};
        D.ain[cell] = grid_op_node_adjoint(D, in, i, j, k, g);
    } else restore_ahead_slot(D, nx, ck, ((int)blockIdx.x - half) * 4 + (int)(threadIdx.x >> 6));
}
// the restore alone (tail reduction: the reduction half is done by the fused particle launch): rides in k_contact_grad's launch, or - a frame without hits -
// runs as this launch; its last D.fk_ride workgroups: forward_kinematics.grad
template <class R>
__global__ __launch_bounds__(BLOCK) void k_restore_ahead(DevSim<R> D, GridSet<R> nx, const Vec4<R>* ck, int restore) {
    if (D.fk_ride > 0 && blockIdx.x >= gridDim.x - D.fk_ride) {
        const int prim = (int)(blockIdx.x - (gridDim.x - D.fk_ride));
        if (threadIdx.x < 13) prim_fk_step_grad(D.prim_state + prim * D.fk_stride, D.prim_grad + prim * D.fk_stride, D.cur_frame, D.dt64, (int)threadIdx.x);
        return;
    }
    if (restore) restore_ahead_slot(D, nx, ck, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
}

// Adjoint of grid_op_mixed4 / mixed3 / mixed2 for the listed particles.  One hit = one group of 32 lanes:
// every lane of the group recomputes the (cheap) shared forward quantities, and the 19 forward-mode
// directions of collide_mixed's adjoint (p_pos3, p_v3, state13) run in 19 different lanes instead of 19
// times in one lane.  Lane n < 27 then owns stencil node n for the mixed2 scatter and the weight adjoints.
// DIRECT: grid_v_mixed.grad contributions go through grid_op_mixed1's node adjoint into grid_v_in.grad (see k_reduce_grid_grad)
// CLOTH: 24 forward-mode directions of the cloth primitive's collide_mixed (p_pos3, p_v3, the face's vertex positions 9 and velocities 9)
// ride (tail reduction: no reduction launch to carry them): the first `ride_blocks` workgroups restore the next frame's forward grid into the other buffer
// set (restore_ahead_slot), the last D.fk_ride workgroups run forward_kinematics.grad of this substep; the hit walk uses the workgroups in between
template <class R, bool DIRECT, bool CLOTH>
__global__ __launch_bounds__(BLOCK) void k_contact_grad(DevSim<R> D, int f, GridSet<R> nx, const Vec4<R>* ck, int ride_blocks) {
    if (D.fk_ride > 0 && blockIdx.x >= gridDim.x - D.fk_ride) {
        const int prim = (int)(blockIdx.x - (gridDim.x - D.fk_ride));
        if (threadIdx.x < 13) prim_fk_step_grad(D.prim_state + prim * D.fk_stride, D.prim_grad + prim * D.fk_stride, D.cur_frame, D.dt64, (int)threadIdx.x);
        return;
    }
    if ((int)blockIdx.x < ride_blocks) {
        restore_ahead_slot(D, nx, ck, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
        return;
    }
    const int wg = (int)blockIdx.x - ride_blocks, nwg = (int)gridDim.x - ride_blocks - (D.fk_ride > 0 ? D.fk_ride : 0);
    __shared__ double pg_acc[MAX_PRIMS * 13];         // primitive-state adjoints of this workgroup's hits (see k_contact)
    // grid_v_mixed.grad corrections of this workgroup's 8 hits are pre-reduced in an LDS tile over the block of its first
    // hit (k_p2g appends a wave's hits contiguously, so the 8 nearly always share the block): neighbouring contact
    // particles hit the same few nodes and their global atomics would serialise on those cache lines
    __shared__ tile_t atile[3 * TILE_WORDS];
    __shared__ R fl_val[BLOCK * 4];                     // the flush's staging (flush_nodes_by_lanes)
    __shared__ unsigned fl_cell[BLOCK];
    __shared__ double jac[MAX_PRIMS * 4 * BLOCK];      // per primitive and lane: d(velocity out)/d(direction of the lane) and the ext_f adjoint's share (each lane reads only what it wrote)
    SMAC_WAVE_T0();
    SMAC_PHASE(32, wg * 8 < 1024);                // (entry; the first 128 workgroups hold hits in the bench scene)
    constexpr int HPW = SMAC_HITS_PER_WAVE, WG_HITS = (BLOCK / 64) * HPW;      // (see k_contact_hits)
    const int grp = HPW == 2 ? (int)(threadIdx.x >> 5) : (int)(threadIdx.x >> 6), d = threadIdx.x & 31, lane0 = (threadIdx.x & 63) & ~31;
    const bool live = HPW == 2 || (threadIdx.x & 32) == 0;
    // a latency chain on an idle chip, as k_contact_hits: hit count, first hit's block and own hit record are asked for together (the records speculatively)
    const int base0 = wg * WG_HITS;
    Hit h_next = D.hits[min(base0 + grp, D.Npad - 1)];
    int block_next = D.hits[min(base0, D.Npad - 1)].block;
    const int nh = *D.nhits;
    if (threadIdx.x < MAX_PRIMS * 13) pg_acc[threadIdx.x] = 0.0;
    __syncthreads();                              // (waits for the three loads as well: the walk needs them at once anyway)
    SMAC_PHASE(33, wg * 8 < 1024);                // hit count, block and record in; accumulators zeroed
    SMAC_WAVE_MARK(1);
    for (int base = base0; base < nh; base += nwg * WG_HITS) {
        for (int i = threadIdx.x; i < 3 * TILE_WORDS; i += BLOCK) atile[i] = 0.0;             // (published by the barrier behind the chains: no load is in flight there)
        const int wg_block = block_next;
        const int hi = base + grp;
        Hit h = {0, 0, 0, 0};
        if (hi < nh && live) h = h_next;
        {
            const int nb2 = base + nwg * WG_HITS;                                             // (a workgroup with a second round: the same, one round ahead)
            if (nb2 < nh) { h_next = D.hits[min(nb2 + grp, D.Npad - 1)]; block_next = D.hits[min(nb2, D.Npad - 1)].block; }
        }
        const int mask = h.mask, p = h.p;
        typename pos_of<R>::type x[3] = {pos_mid<R>(), pos_mid<R>(), pos_mid<R>()};
        if (mask) load_pos(frame(D.S, f, D.Npad), D.Npad, p, x);
        // asked for NOW, used at the end: x.grad's old value (lane d < 3) and - DIRECT - the {m,p} records of the tile nodes this thread flushes.  Read where they are
        // used, each would wait behind this wave's global atomics (loads and atomics retire in order) or add a round trip of its own
        R ax_old = R(0);
        if (mask && d < 3) ax_old = D.Af[rowoff(CX + d, p, D.Npad)];
        static_assert(TILE_WORDS <= BLOCK, "one tile node per thread in the flush below");
        Vec4<R> flush_in = {R(0), R(0), R(0), R(0)};
        unsigned flush_cell = 0u;
        int flush_i = 0, flush_j = 0, flush_k = 0;
        if (threadIdx.x < TILE_WORDS) {
            const int nb = D.nb, idx = threadIdx.x;
            flush_i = 4 * (wg_block / (nb * nb)) + idx / TSX;
            flush_j = 4 * ((wg_block / nb) % nb) + (idx / TSY) % TW;
            flush_k = 4 * (wg_block % nb) + idx % TW;
            if (flush_i < D.n && flush_j < D.n && flush_k < D.n) {                              // (the tile of a block at the far wall overhangs the grid; nothing is added there)
                flush_cell = (unsigned)cell_of(nb, flush_i, flush_j, flush_k);
                if (DIRECT) flush_in = gld(D.vin, flush_cell);
            }
        }
        SMAC_PHASE(34, wg * 8 < 1024 && base == base0);        // tile zeroed, position / x.grad / flush records asked for
        Stencil<R> st;
        Nodes nd;
        stencil_at(D, x, st, nd, h.block);
        const double life = 1.0 / (double)(D.substeps - (f - D.frame_shift) % D.substeps);
        const double x64[3] = {pos_get(x[0]), pos_get(x[1]), pos_get(x[2])};
        const double pm64 = (double)D.p_mass;
        // this lane's stencil node (lanes 27..31 idle in the node-parallel parts)
        const int n = d < 27 ? d : 0;
        const int ni = n / 9, nj = (n / 3) % 3, nk = n % 3;
        const R wx = ni == 0 ? st.w[0][0] : (ni == 1 ? st.w[1][0] : st.w[2][0]);
        const R wy = nj == 0 ? st.w[0][1] : (nj == 1 ? st.w[1][1] : st.w[2][1]);
        const R wz = nk == 0 ? st.w[0][2] : (nk == 1 ? st.w[1][2] : st.w[2][2]);
        const R wn = d < 27 ? wx * wy * wz : R(0);
        const unsigned cell = (unsigned)((ni == 0 ? nd.cx[0] : (ni == 1 ? nd.cx[1] : nd.cx[2])) + (nj == 0 ? nd.cy[0] : (nj == 1 ? nd.cy[1] : nd.cy[2])) +
                                         (nk == 0 ? nd.cz[0] : (nk == 1 ? nd.cz[1] : nd.cz[2])));
        // node-parallel gathers, group-reduced (xor shuffles stay inside the 32-lane group)
        Vec4<R> vm = {R(0), R(0), R(0), R(0)}, G = {R(0), R(0), R(0), R(0)}, vin_n = {R(0), R(0), R(0), R(0)};
        R has = R(0);
        if (mask && d < 27) {
            G = gld(D.aout, cell);
            vin_n = gld(D.vin, cell);
            vm = grid_v_mixed_at(D, vin_n, cell);
            has = vin_n.x > D.m_eps ? R(1) : R(0);
        }
        R v_tmp[3] = {wn * vm.x, wn * vm.y, wn * vm.z};                                     // mixed2 forward
        R gd[3] = {-R(2) * wn * has * G.x, -R(2) * wn * has * G.y, -R(2) * wn * has * G.z};   // mixed4.grad: d/d(v_tmp - v_tgt)
#pragma unroll
        for (int o = 16; o > 0; o >>= 1)
#pragma unroll
            for (int c = 0; c < 3; ++c) { v_tmp[c] += __shfl_xor(v_tmp[c], o, 64); gd[c] += __shfl_xor(gd[c], o, 64); }
        // forward chain (every lane), then its adjoint primitive by primitive in reverse - in double whatever R is (the
        // push-out and its derivative carry a factor 1/dt)
        SMAC_WAVE_MARK(2);
        SMAC_WAVE_NOTE((__shfl(mask, 0, 64) & 0xff) | ((__shfl(mask, 32, 64) & 0xff) << 8));
        SMAC_PHASE(35, wg * 8 < 1024 && base == base0);        // position in, 27 node records gathered and reduced over the group
        constexpr bool HYB = SMAC_CONTACT_HYBRID && sizeof(R) == 4;          // (see k_contact_hits: float32 mode keeps only the distance in double)
        // (CLOTH instantiation: sheet bits | primitive band bits << 8, see k_contact_hits; the forward chain is primitives in index order, then the sheet,
        //  so the adjoint takes the sheet first and the primitives in reverse)
        const int pm = CLOTH ? (mask >> 8) & 15 : mask;
        const bool cloth_on = CLOTH && (mask & 1) != 0;
        // FORWARD over the primitives in range, in index order, ONE dual pass each: its value part is the forward velocity the next primitive starts from, its
        // tangent part - lane d: direction d - is filed per lane (jac: the velocity's three tangents and the ext_f adjoint's share, which does not depend on the
        // velocity adjoint) for the reverse walk below.  Rounds 1-4 walked in reverse and re-ran the plain chain up to every primitive: a particle between both fingers
        // cost 3 plain + 2 dual chains, and the launch lasts as long as its slowest wave (profiles/r05_contact_grid.txt); now 2 dual chains, nothing replayed.
        double v_tgt[3] = {(double)v_tmp[0], (double)v_tmp[1], (double)v_tmp[2]};
#pragma unroll 1
        for (int i = 0; i < D.P; ++i) {
            const bool act = (pm >> i) & 1;
            if (!__ballot(act)) continue;
            double jv[3] = {0.0, 0.0, 0.0}, je = 0.0, vfwd[3] = {0.0, 0.0, 0.0};
            if (act && d < 19) {
                const double* ps = D.prim_state + ((size_t)i * D.max_frames + f) * 13;
                Dual<double> pos[3], stt[13];
                for (int c = 0; c < 3; ++c) pos[c] = Dual<double>(x64[c], d == c ? 1.0 : 0.0);
                for (int c = 0; c < 13; ++c) stt[c] = Dual<double>(ps[c], d == 6 + c ? 1.0 : 0.0);
                if (HYB) {
                    Dual<float> v[3], ext[6];
                    for (int c = 0; c < 3; ++c) v[c] = Dual<float>((float)v_tgt[c], d == 3 + c ? 1.f : 0.f);
                    collide_mixed_hybrid<Dual<double>, Dual<float>>(D.prim64[i], *(const PrimTable<float>*)&D.prim[i], stt, pos, v, (float)D.p_mass, D.dt64, life, ext);
                    for (int c = 0; c < 3; ++c) { jv[c] = (double)v[c].d; vfwd[c] = (double)v[c].v; }
                    for (int c = 0; c < 6; ++c) je += D.ext_f_grad[i * 6 + c] * (double)ext[c].d;
                } else {
                    Dual<double> v[3], ext[6];
                    for (int c = 0; c < 3; ++c) v[c] = Dual<double>(v_tgt[c], d == 3 + c ? 1.0 : 0.0);
                    collide_mixed(D.prim64[i], stt, pos, v, pm64, D.dt64, life, ext);
                    for (int c = 0; c < 3; ++c) { jv[c] = v[c].d; vfwd[c] = v[c].v; }
                    for (int c = 0; c < 6; ++c) je += D.ext_f_grad[i * 6 + c] * ext[c].d;
                }
            }
            jac[(i * 4 + 0) * BLOCK + threadIdx.x] = jv[0]; jac[(i * 4 + 1) * BLOCK + threadIdx.x] = jv[1];
            jac[(i * 4 + 2) * BLOCK + threadIdx.x] = jv[2]; jac[(i * 4 + 3) * BLOCK + threadIdx.x] = je;
            // the velocity this primitive leaves, from the group's first lane (outside the band the chain leaves it as it was)
            const double f0 = __shfl(vfwd[0], lane0, 64), f1 = __shfl(vfwd[1], lane0, 64), f2 = __shfl(vfwd[2], lane0, 64);
            if (act) { v_tgt[0] = f0; v_tgt[1] = f1; v_tgt[2] = f2; }
        }
        double g[3] = {-(double)gd[0], -(double)gd[1], -(double)gd[2]};             // adjoint of v_tgt
        double gpos[3] = {0.0, 0.0, 0.0};
        if (CLOTH) {
            const ClothDev& Cl = D.cloth;
            const double sc = Cl.par.scale;
            const bool act = cloth_on;
            const int* vid = Cl.faces + 3 * (act ? h.pad : 0);
            double out = 0.0, vfwd[3] = {0.0, 0.0, 0.0};
            if (act && d < 24) {
                // particle position / velocity are unit-domain variables (x = sc * u), the vertices physical; the velocity is the one the primitives left
                Dual<double> pos[3], v[3], xv[3][3], vv[3][3], cf[3], wb[3];
                for (int c = 0; c < 3; ++c) {
                    pos[c] = Dual<double>(sc * x64[c], d == c ? sc : 0.0);
                    v[c] = Dual<double>(sc * v_tgt[c], d == 3 + c ? sc : 0.0);
                }
                for (int i = 0; i < 3; ++i)
                    for (int c = 0; c < 3; ++c) {
                        xv[i][c] = Dual<double>(Cl.pos[((size_t)f * Cl.V + vid[i]) * 3 + c], d == 6 + 3 * i + c ? 1.0 : 0.0);
                        vv[i][c] = Dual<double>(Cl.vel[((size_t)f * Cl.V + vid[i]) * 3 + c], d == 15 + 3 * i + c ? 1.0 : 0.0);
                    }
                const bool in_band = cloth_collide_mixed<Dual<double>>(Cl.par, xv, vv, pos, v, pm64 * sc * sc, D.dt64, life, (mask >> 1) & 1, cf, wb);
                for (int c = 0; c < 3; ++c) { out += g[c] * v[c].d / sc; vfwd[c] = v[c].v / sc; }      // (outside the band v is untouched: identity)
                if (in_band)
                    for (int i = 0; i < 3; ++i)
                        for (int c = 0; c < 3; ++c) out += Cl.ext_f_grad[(size_t)vid[i] * 3 + c] * (cf[c] * wb[i]).d;
            }
            const double f0 = __shfl(vfwd[0], lane0, 64), f1 = __shfl(vfwd[1], lane0, 64), f2 = __shfl(vfwd[2], lane0, 64);
            if (act) { v_tgt[0] = f0; v_tgt[1] = f1; v_tgt[2] = f2; }
            const double o0 = __shfl(out, lane0 + 0, 64), o1 = __shfl(out, lane0 + 1, 64), o2 = __shfl(out, lane0 + 2, 64);
            const double o3 = __shfl(out, lane0 + 3, 64), o4 = __shfl(out, lane0 + 4, 64), o5 = __shfl(out, lane0 + 5, 64);
            if (act) {
                gpos[0] += o0; gpos[1] += o1; gpos[2] += o2;
                g[0] = o3; g[1] = o4; g[2] = o5;
                if (d >= 6 && d < 24 && out != 0.0 && Cl.pos_grad) {                // position.grad[f, vertex] / velocity.grad[f, vertex]
                    const int q = d < 15 ? d - 6 : d - 15, vi = q / 3, c = q % 3;
                    double* dst = (d < 15 ? Cl.pos_grad : Cl.vel_grad) + ((size_t)f * Cl.V + (vi == 0 ? vid[0] : (vi == 1 ? vid[1] : vid[2]))) * 3 + c;
                    atomic_add(dst, out);
                }
            }
        }
#pragma unroll 1
        for (int i = D.P - 1; i >= 0; --i) {                 // REVERSE: the filed tangents against the velocity adjoint of the moment
            const bool act = (pm >> i) & 1;
            if (!__ballot(act)) continue;
            double out = 0.0;
            if (act && d < 19)
                out = g[0] * jac[(i * 4 + 0) * BLOCK + threadIdx.x] + g[1] * jac[(i * 4 + 1) * BLOCK + threadIdx.x] + g[2] * jac[(i * 4 + 2) * BLOCK + threadIdx.x] +
                      jac[(i * 4 + 3) * BLOCK + threadIdx.x];
            // direction d of this group's hit sits in lane lane0 + d
            const double o0 = __shfl(out, lane0 + 0, 64), o1 = __shfl(out, lane0 + 1, 64), o2 = __shfl(out, lane0 + 2, 64);
            const double o3 = __shfl(out, lane0 + 3, 64), o4 = __shfl(out, lane0 + 4, 64), o5 = __shfl(out, lane0 + 5, 64);
            if (act) {
                gpos[0] += o0; gpos[1] += o1; gpos[2] += o2;
                g[0] = o3; g[1] = o4; g[2] = o5;
            }
            // state adjoint: lanes 6..18 hold component d-6; sum the two groups of the wave, one atomic per wave
            double sg = (act && d >= 6 && d < 19) ? out : 0.0;
            sg += __shfl_xor(sg, 32, 64);
            if ((threadIdx.x & 63) >= 6 && (threadIdx.x & 63) < 19 && sg != 0.0)
                __hip_atomic_fetch_add(pg_acc + i * 13 + (d - 6), sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        SMAC_WAVE_MARK(3);
        SMAC_PHASE(36, wg * 8 < 1024 && base == base0);        // forward replay + dual chains of the primitives in range
        const R diff[3] = {(R)((double)v_tmp[0] - v_tgt[0]), (R)((double)v_tmp[1] - v_tgt[1]), (R)((double)v_tmp[2] - v_tgt[2])};
        __syncthreads();                                                                        // tile zeroed everywhere
        R late[4] = {R(0), R(0), R(0), R(0)};                  // a node outside the tile: its global atomics leave AFTER the barrier below (queued adds hold up their
        bool is_late = false;                                  // wave's next memory instruction, and with the wave everybody at a barrier)
        if (mask) {
            // adjoint of v_tmp = direct (mixed4) + through the chain (mixed3); mixed2.grad scatter by node
            const R gvt[3] = {(R)((double)gd[0] + g[0]), (R)((double)gd[1] + g[1]), (R)((double)gd[2] + g[2])};
            R gw = R(0);
            if (d < 27) {
                const bool in_tile = h.block == wg_block && ((nd.okx >> ni) & (nd.oky >> nj) & (nd.okz >> nk) & 1) != 0;
                const int tw = (ni == 0 ? nd.tx[0] : (ni == 1 ? nd.tx[1] : nd.tx[2])) + (nj == 0 ? nd.ty[0] : (nj == 1 ? nd.ty[1] : nd.ty[2])) +
                               (nk == 0 ? nd.tz[0] : (nk == 1 ? nd.tz[1] : nd.tz[2]));
                if (DIRECT && !in_tile) {
                    int ci, cj, ck;
                    cell_ijk(D.nb, cell, ci, cj, ck);
                    R gn[3] = {wn * gvt[0], wn * gvt[1], wn * gvt[2]};
                    const Vec4<R> o = grid_op_node_adjoint(D, vin_n, ci, cj, ck, gn);
                    if (has != R(0)) { is_late = true; late[0] = o.x; late[1] = o.y; late[2] = o.z; late[3] = o.w; }
                } else if (in_tile) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) lds_add(atile + tw + c * TILE_WORDS, wn * gvt[c]);
                } else {
                    is_late = true;
#pragma unroll
                    for (int c = 0; c < 3; ++c) late[c] = wn * gvt[c];
                }
                gw = vm.x * gvt[0] + vm.y * gvt[1] + vm.z * gvt[2]                           // mixed2: d/dw
                     - R(2) * has * (diff[0] * G.x + diff[1] * G.y + diff[2] * G.z);        // mixed4: d/dw
            }
            // weight adjoint -> fx adjoint: this node contributes to one factor per dimension
            R gfx[3];
            gfx[0] = gw * wy * wz * (ni == 0 ? st.dw[0][0] : (ni == 1 ? st.dw[1][0] : st.dw[2][0]));
            gfx[1] = gw * wx * wz * (nj == 0 ? st.dw[0][1] : (nj == 1 ? st.dw[1][1] : st.dw[2][1]));
            gfx[2] = gw * wx * wy * (nk == 0 ? st.dw[0][2] : (nk == 1 ? st.dw[1][2] : st.dw[2][2]));
            if (d >= 27) gfx[0] = gfx[1] = gfx[2] = R(0);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1)
#pragma unroll
                for (int c = 0; c < 3; ++c) gfx[c] += __shfl_xor(gfx[c], o, 64);
            if (d < 3) {
                const double gp = d == 0 ? gpos[0] : (d == 1 ? gpos[1] : gpos[2]);
                const R gf = d == 0 ? gfx[0] : (d == 1 ? gfx[1] : gfx[2]);
                D.Af[rowoff(CX + d, p, D.Npad)] = ax_old + (R)(gp + (double)(D.inv_dx * gf));   // (a particle stands in the list once: nobody else adds to this word)
            }
        }
        SMAC_PHASE(37, wg * 8 < 1024 && base == base0);        // node scatter into the LDS tile, x.grad out
        __syncthreads();
        SMAC_WAVE_MARK(4);
        SMAC_PHASE(38, wg * 8 < 1024 && base == base0);        // barrier
        if (__syncthreads_or(is_late)) {                       // (hits of another block than the tile's: all 27 nodes of such a hit come this way - by lanes as the tile's)
            const Vec4<R> lo = {late[0], late[1], late[2], late[3]};
            flush_nodes_by_lanes(DIRECT ? D.ain : D.amix, lo, cell, fl_val, fl_cell);
        }
        {
            Vec4<R> o = {R(0), R(0), R(0), R(0)};
            if (threadIdx.x < TILE_WORDS) {
                const int idx = threadIdx.x;
                const R a0 = (R)atile[idx], a1 = (R)atile[TILE_WORDS + idx], a2 = (R)atile[2 * TILE_WORDS + idx];
                if (a0 != R(0) || a1 != R(0) || a2 != R(0)) {
                    R gn[3] = {a0, a1, a2};
                    if (DIRECT) o = grid_op_node_adjoint(D, flush_in, flush_i, flush_j, flush_k, gn);
                    else o = Vec4<R>{a0, a1, a2, R(0)};
                }
            }
            flush_nodes_by_lanes(DIRECT ? D.ain : D.amix, o, flush_cell, fl_val, fl_cell);     // (its closing barrier: the tile may be zeroed again)
        }
        SMAC_WAVE_MARK(5);
        SMAC_PHASE(39, wg * 8 < 1024 && base == base0);        // late atomics, tile flushed through grid_op's node adjoint (global atomics)
    }
    __syncthreads();
    if (threadIdx.x < D.P * 13 && pg_acc[threadIdx.x] != 0.0) {
        const int i = threadIdx.x / 13, c = threadIdx.x % 13;
        atomic_add(D.prim_grad + ((size_t)i * D.max_frames + f) * 13 + c, pg_acc[threadIdx.x]);
    }
    SMAC_PHASE(40, wg * 8 < 1024);
    SMAC_WAVE_HIST(1);
    SMAC_WAVE_SLOW(36000ull, (unsigned long long)wg | ((unsigned long long)(threadIdx.x >> 6) << 16), nh);
}

// Adjoint of the penalty contact impulse of p2g (collision_type 1) for the listed particles: the impulse's
// adjoint is sum_nodes w * grid_v_in.grad (what p2g.grad calls gvp); 32 lanes per hit, lane n < 27 gathers node n,
// lane d < 19 runs forward-mode direction d (p_pos3, p_v3, state13).
// CLOTH: the sheet's collide_particle, 24 directions (p_pos3, p_v3, vertex positions 9, vertex velocities 9)
template <class R, bool CLOTH>
__global__ __launch_bounds__(BLOCK) void k_particle_contact_grad(DevSim<R> D, int f) {
    __shared__ double pg_acc[MAX_PRIMS * 13];         // primitive-state adjoints of this workgroup's hits (see k_contact)
    if (threadIdx.x < MAX_PRIMS * 13) pg_acc[threadIdx.x] = 0.0;
    __syncthreads();
    const int nh = *D.nhits;
    const int grp = threadIdx.x >> 5, d = threadIdx.x & 31, lane0 = (threadIdx.x & 63) & ~31;
    for (int base = blockIdx.x * (BLOCK / 32); base < nh; base += gridDim.x * (BLOCK / 32)) {
        const int hi = base + grp;
        Hit h = {0, 0, 0, 0};
        if (hi < nh) h = D.hits[hi];
        const int mask = h.mask, p = h.p;
        typename pos_of<R>::type xp[3] = {pos_mid<R>(), pos_mid<R>(), pos_mid<R>()};
        R v[3] = {R(0), R(0), R(0)};
        if (mask) {
            load_pos(frame(D.S, f, D.Npad), D.Npad, p, xp);
            load_vec(frame(D.S, f, D.Npad), CV, 3, D.Npad, p, v);
        }
        const double x64[3] = {pos_get(xp[0]), pos_get(xp[1]), pos_get(xp[2])};
        Stencil<R> st;
        Nodes nd;
        stencil_at(D, xp, st, nd, h.block);
        const int n = d < 27 ? d : 0;
        const int ni = n / 9, nj = (n / 3) % 3, nk = n % 3;
        const R wn = (ni == 0 ? st.w[0][0] : (ni == 1 ? st.w[1][0] : st.w[2][0])) * (nj == 0 ? st.w[0][1] : (nj == 1 ? st.w[1][1] : st.w[2][1])) *
                     (nk == 0 ? st.w[0][2] : (nk == 1 ? st.w[1][2] : st.w[2][2]));
        const unsigned cell = (unsigned)((ni == 0 ? nd.cx[0] : (ni == 1 ? nd.cx[1] : nd.cx[2])) + (nj == 0 ? nd.cy[0] : (nj == 1 ? nd.cy[1] : nd.cy[2])) +
                                         (nk == 0 ? nd.cz[0] : (nk == 1 ? nd.cz[1] : nd.cz[2])));
        R gi[3] = {R(0), R(0), R(0)};
        if (mask && d < 27) {
            const Vec4<R> a = gld(D.ain, cell);
            gi[0] = wn * a.y; gi[1] = wn * a.z; gi[2] = wn * a.w;
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1)
#pragma unroll
            for (int c = 0; c < 3; ++c) gi[c] += __shfl_xor(gi[c], o, 64);
        if (CLOTH) {
            const ClothDev& Cl = D.cloth;
            const double sc = Cl.par.scale;
            const bool act = mask != 0;
            const int* vid = Cl.faces + 3 * (act ? h.pad : 0);
            double out = 0.0;
            if (act && d < 24) {
                Dual<double> xs[3], vs[3], xv[3][3], vv[3][3], im[3], cf[3], wb[3];
                for (int c = 0; c < 3; ++c) {
                    xs[c] = Dual<double>(sc * x64[c], d == c ? sc : 0.0);
                    vs[c] = Dual<double>(sc * (double)v[c], d == 3 + c ? sc : 0.0);
                }
                for (int i = 0; i < 3; ++i)
                    for (int c = 0; c < 3; ++c) {
                        xv[i][c] = Dual<double>(Cl.pos[((size_t)f * Cl.V + vid[i]) * 3 + c], d == 6 + 3 * i + c ? 1.0 : 0.0);
                        vv[i][c] = Dual<double>(Cl.vel[((size_t)f * Cl.V + vid[i]) * 3 + c], d == 15 + 3 * i + c ? 1.0 : 0.0);
                    }
                if (cloth_collide_particle<Dual<double>>(Cl.par, xv, vv, xs, vs, D.dt64, (mask >> 1) & 1, im, cf, wb)) {
                    for (int c = 0; c < 3; ++c) out += (double)gi[c] * im[c].d / (sc * sc * sc);
                    for (int i = 0; i < 3; ++i)
                        for (int c = 0; c < 3; ++c) out += Cl.ext_f_grad[(size_t)vid[i] * 3 + c] * (cf[c] * wb[i]).d;
                }
            }
            if (act && d < 6) {
                R* Af = D.Af;
                Af[rowoff((d < 3 ? CX : CV - 3) + d, p, D.Npad)] += (R)out;
            }
            if (act && d >= 6 && d < 24 && out != 0.0 && Cl.pos_grad) {
                const int q = d < 15 ? d - 6 : d - 15, vi = q / 3, c = q % 3;
                atomic_add((d < 15 ? Cl.pos_grad : Cl.vel_grad) + ((size_t)f * Cl.V + (vi == 0 ? vid[0] : (vi == 1 ? vid[1] : vid[2]))) * 3 + c, out);
            }
        }
#pragma unroll 1
        for (int i = 0; i < (CLOTH ? 0 : D.P); ++i) {
            const bool act = (mask >> i) & 1;
            if (!__ballot(act)) continue;
            double out = 0.0;
            if (act && d < 19) {                                   // f64 duals (see k_p2g: the impulse scales with a small penetration depth)
                const double* ps = D.prim_state + ((size_t)i * D.max_frames + f) * 13;
                Dual<double> xs[3], vs[3], ss[13], im[3], es[6];
                for (int c = 0; c < 3; ++c) { xs[c] = Dual<double>(x64[c], d == c ? 1.0 : 0.0); vs[c] = Dual<double>((double)v[c], d == 3 + c ? 1.0 : 0.0); }
                for (int c = 0; c < 13; ++c) ss[c] = Dual<double>(ps[c], d == 6 + c ? 1.0 : 0.0);
                if (collide_particle(D.prim64[i], ss, xs, vs, D.dt64, im, es)) {
                    for (int c = 0; c < 3; ++c) out += (double)gi[c] * im[c].d;
                    for (int c = 0; c < 6; ++c) out += D.ext_f_grad[i * 6 + c] * es[c].d;
                }
            }
            if (act && d < 6) {
                R* Af = D.Af;
                Af[rowoff((d < 3 ? CX : CV - 3) + d, p, D.Npad)] += (R)out;
            }
            double sg = (act && d >= 6 && d < 19) ? out : 0.0;
            sg += __shfl_xor(sg, 32, 64);
            if ((threadIdx.x & 63) >= 6 && (threadIdx.x & 63) < 19 && sg != 0.0)
                __hip_atomic_fetch_add(pg_acc + i * 13 + (d - 6), sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            (void)lane0;
        }
    }
    __syncthreads();
    if (threadIdx.x < D.P * 13 && pg_acc[threadIdx.x] != 0.0) {
        const int i = threadIdx.x / 13, c = threadIdx.x % 13;
        atomic_add(D.prim_grad + ((size_t)i * D.max_frames + f) * 13 + c, pg_acc[threadIdx.x]);
    }
}

template <class R, bool GRIDC, bool HALO = false>
__global__ __launch_bounds__(BLOCK) void k_grid_op_grad(DevSim<R> D) {
    int b, l, i, j, k;
    size_t cell;
    if (!active_cell(D, b, l, cell, i, j, k)) return;
    const Vec4<R> in = D.vin[cell];
    Vec4<R> go = D.aout[cell];
    if (HALO && D.halo_hs.count) {                                       // the neighbours' partial sums of grid_v_out.grad on the shared planes (see k_reduce_aout)
        const size_t total = (size_t)D.halo_np * D.n * D.n;
        for (int s = 0; s < D.halo_hs.count; ++s) {
            const int pi = i - D.halo_hs.plane0[s];
            if ((unsigned)pi < (unsigned)D.halo_np) {
                const Vec4<R> a = D.halo_recv[(size_t)D.halo_hs.slot[s] * total + ((size_t)pi * D.n + j) * D.n + k];
                go.x += a.x; go.y += a.y; go.z += a.z; go.w += a.w;
            }
        }
    }
    const Vec4<R> zero4 = {R(0), R(0), R(0), R(0)};
    Vec4<R> gm_ = zero4;
    if (D.collision_type == CONTACT_MIXED) gm_ = D.amix[cell];
    const R m = in.x;
    if (!(m > D.m_eps)) { D.ain[cell] = zero4; return; }     // (written, so that grid_v_in.grad needs no zeroing in front of this kernel: the slab pieces' fused backward launch)
    const R inv = R(1) / m;
    const R vin[3] = {in.y, in.z, in.w};
    R g[3] = {go.x + gm_.x, go.y + gm_.y, go.z + gm_.z};                               // grid_v_out += grid_v_mixed
    R v[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) v[d] = inv * vin[d] + D.dt * D.g[d];
    R gm_extra = R(0);
    constexpr bool gridc = GRIDC;
    R v0[3] = {v[0], v[1], v[2]};
    if (gridc) {                                    // forward chain again: the boundary acts on the post-contact velocity
        const R pos[3] = {R(i) * D.dx, R(j) * D.dx, R(k) * D.dx};
        R dummy[6];
        for (int q = 0; q < D.P; ++q) {
            if (!D.prim[q].contact) continue;
            R s13[13];
            prim_state_R(D, q, D.cur_frame, s13);
            collide_grid(D.prim[q], s13, pos, v, m, D.dt, dummy);
        }
    }
    const int mask = boundary(D, i, j, k, v);
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (mask & (1 << d)) g[d] = R(0);
    if (gridc) {                                    // adjoint of the chain, last primitive first (17 forward-mode directions)
        const R pos[3] = {R(i) * D.dx, R(j) * D.dx, R(k) * D.dx};
#pragma unroll 1
        for (int q = D.P - 1; q >= 0; --q) {
            if (!D.prim[q].contact) continue;
            R s13[13], vq[3] = {v0[0], v0[1], v0[2]}, dummy[6];
            prim_state_R(D, q, D.cur_frame, s13);
            for (int r = 0; r < q; ++r) {           // velocity entering primitive q
                if (!D.prim[r].contact) continue;
                R sr[13];
                prim_state_R(D, r, D.cur_frame, sr);
                collide_grid(D.prim[r], sr, pos, vq, m, D.dt, dummy);
            }
            R probe[3] = {vq[0], vq[1], vq[2]};
            if (!collide_grid(D.prim[q], s13, pos, probe, m, D.dt, dummy)) continue;       // identity for this node
            R out[17];
            for (int dir = 0; dir < 17; ++dir) {
                Dual<R> vs[3], ss[13], es[6];
                for (int c = 0; c < 3; ++c) vs[c] = Dual<R>(vq[c], dir == c ? R(1) : R(0));
                for (int c = 0; c < 13; ++c) ss[c] = Dual<R>(s13[c], dir == 4 + c ? R(1) : R(0));
                collide_grid(D.prim[q], ss, pos, vs, Dual<R>(m, dir == 3 ? R(1) : R(0)), D.dt, es);
                R acc = R(0);
                for (int c = 0; c < 3; ++c) acc += g[c] * vs[c].d;
                for (int c = 0; c < 6; ++c) acc += (R)D.ext_f_grad[q * 6 + c] * es[c].d;
                out[dir] = acc;
            }
            g[0] = out[0]; g[1] = out[1]; g[2] = out[2];
            gm_extra += out[3];
            for (int c = 0; c < 13; ++c) atomic_add(D.prim_grad + ((size_t)q * D.max_frames + D.cur_frame) * 13 + c, (double)out[4 + c]);
        }
    }
    R gm = R(0);
#pragma unroll
    for (int d = 0; d < 3; ++d) gm -= vin[d] * g[d];
    const Vec4<R> o = {gm * inv * inv + gm_extra, g[0] * inv, g[1] * inv, g[2] * inv};
    D.ain[cell] = o;
}

// Register budget: the unrolled 27-node gather, the SVD factors and the adjoint accumulators do not fit
// 128 VGPRs together, and at 1 wave/SIMD the kernel is latency-bound.  The SVD factors (kept for the
// constitutive adjoint) are parked in LDS across the gather loop, and the loop is fenced per x-plane so
// that at most 9 nodes of loads are in flight: 4 waves/SIMD instead of 1.
#ifndef SMAC_STASH21
#define SMAC_STASH21 SMAC_WIDE_TILE   // park U, V, e only - F_tmp - I comes back from the C, E rows the F_tmp adjoint reloads anyway, e' and J - 1 from e and F_tmp: 21 slots.
#endif                           // +1.5 us on the fused kernel by itself (profiles/r04_w_occupancy4.txt); it is what lets the wide gather / scatter tiles fit 3 workgroups per CU
constexpr int STASH = SMAC_STASH21 ? 21 : 34;   // U9 V9 e3 ep3 Et9 Jm1: parked in LDS across the gather loop
template <class R> struct occ { static constexpr int heavy = sizeof(R) == 4 ? SMAC_OCC_P2GG : 2; };   // waves/SIMD asked of the register allocator
// p2g.grad + svd_grad + compute_F_tmp.grad of one particle (mpm_simulator.py:371-374): `gt` is the staged grid_v_in.grad / grid_m.grad tile, `stash` the
// workgroup's LDS parking space.  Writes the adjoint of frame f to D.Af; KEEP: also hands x.grad, v.grad, C.grad of frame f back in registers
// (k_p2g_g2p_grad feeds them to the G2P adjoint of the substep before) - only with ACC_VCF = false, i.e. when frame f carried no seed.
template <class R, bool ACC_VCF, bool PCON, bool KEEP, bool MAT2 = false>
__device__ __forceinline__ void p2g_grad_particle(const DevSim<R>& D, int f, const Chunk& ch, int p, int t, typename const_t<R>::type* stash, const Vec4<R>* gt,
                                                  R* gx_o, R* gv_o, R* gC_o, int pa) {      // pa: where particle p's rows lie in D.An (p, or An_map[p])
    typedef typename const_t<R>::type CT;
    const R* Sf = frame(D.S, f, D.Npad);
    const R* An = D.An;
    R* Af = D.Af;
    // every global load of the kernel is issued here, in one batch (one memory round trip instead of four);
    // C, E and the SVD factors wait in LDS until the constitutive adjoint needs them
    typename pos_of<R>::type x[3];
    R v[3], aff[9];
    R gFn[9];
    {
        R C[9], E[9];
        CT Et[9], En[9], stress[9];
        load_vec(Sf, CC, 9, D.Npad, p, C);
        load_vec(Sf, CF, 9, D.Npad, p, E);
        load_pos(Sf, D.Npad, p, x);
        load_vec(Sf, CV, 3, D.Npad, p, v);
        load_vec(An, CF, 9, D.Npad, pa, gFn);      // F.grad[f+1]: fetched with the rest
        ConstState<CT> cs;
        {
            CT Cc[9], Ec[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) { Cc[i] = (CT)C[i]; Ec[i] = (CT)E[i]; }
            f_tmp(Cc, Ec, (CT)D.dt, Et);
            const Material<CT> mat = particle_material<CT, MAT2>(D, p);
            constitutive_fwd(mat, Et, En, stress, cs);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) aff[i] = D.stress_scale * (R)stress[i] + D.p_mass * C[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            stash[i * BLOCK + t] = cs.U[i];
            stash[(9 + i) * BLOCK + t] = cs.V[i];
            if (!SMAC_STASH21) stash[(24 + i) * BLOCK + t] = Et[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            stash[(18 + i) * BLOCK + t] = cs.e[i];
            if (!SMAC_STASH21) stash[(21 + i) * BLOCK + t] = cs.ep[i];
        }
        if (!SMAC_STASH21) stash[33 * BLOCK + t] = cs.Jm1;
    }
    SMAC_PHASE(2, KEEP);                       // rows loaded, forward constitutive model recomputed and parked
    R imp[3] = {R(0), R(0), R(0)};
    int ci = -1;
    if (D.n_control > 0) {
        ci = D.control_idx[D.orig_id[p]];
        if (ci >= 0)
            for (int d = 0; d < 3; ++d) imp[d] = R(6e-4) * D.action[3 * ci + d] * D.dt;
    }
    // (PCON is a template parameter for k_p2g only.  Here the instantiation WITH the penalty chain compiled in gets the better
    //  register allocation - 145 VGPRs instead of 168 + spills, 20 us at 1M particles - so every launch uses it and the test is dynamic.)
    if (PCON && D.collision_type == CONTACT_PARTICLE && D.any_contact) {      // the contact impulse is part of the scattered momentum
        const int cm = D.pmask[p];
        if (cm && D.cloth.present) {
            const ClothDev& Cl = D.cloth;
            const double sc = Cl.par.scale;
            const int* vid = Cl.faces + 3 * Cl.contact_id[(size_t)f * Cl.n_ids + D.orig_id[p]];
            double xv[3][3], vv[3][3];
            for (int i = 0; i < 3; ++i)
                for (int c = 0; c < 3; ++c) {
                    xv[i][c] = Cl.pos[((size_t)f * Cl.V + vid[i]) * 3 + c];
                    vv[i][c] = Cl.vel[((size_t)f * Cl.V + vid[i]) * 3 + c];
                }
            const double pp[3] = {sc * pos_get(x[0]), sc * pos_get(x[1]), sc * pos_get(x[2])}, pvv[3] = {sc * (double)v[0], sc * (double)v[1], sc * (double)v[2]};
            double im[3], cf[3], wb[3];
            if (cloth_collide_particle<double>(Cl.par, xv, vv, pp, pvv, D.dt64, (cm >> 1) & 1, im, cf, wb))
                for (int c = 0; c < 3; ++c) imp[c] += (R)(im[c] / (sc * sc * sc));
        } else if (cm) {
            const double x64[3] = {pos_get(x[0]), pos_get(x[1]), pos_get(x[2])}, v64[3] = {(double)v[0], (double)v[1], (double)v[2]};
#pragma unroll 1
            for (int i = 0; i < D.P; ++i) {
                if (!((cm >> i) & 1)) continue;
                const double* ps = D.prim_state + ((size_t)i * D.max_frames + f) * 13;
                double s13[13], im[3], ex[6];
                for (int c = 0; c < 13; ++c) s13[c] = ps[c];
                if (collide_particle(D.prim64[i], s13, x64, v64, D.dt64, im, ex))
                    for (int c = 0; c < 3; ++c) imp[c] += (R)im[c];
            }
        }
    }
    R pv[3] = {D.p_mass * v[0] + imp[0], D.p_mass * v[1] + imp[1], D.p_mass * v[2] + imp[2]};
    Stencil<R> st;
    Nodes nd;
    stencil_at_p(D, x, st, nd, ch.block);
    WGrad<R> wg;
    wg.zero();
    R gvp[3] = {R(0), R(0), R(0)}, gfx[3] = {R(0), R(0), R(0)};
    R gaff[9] = {R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0)};
    const bool all_in = (nd.okx & nd.oky & nd.okz) == 7;
    // Separable form.  With a(n) = {grid_m.grad, grid_v_in.grad} at node n and the scattered momentum affine in
    // the offset, mom(i,j,k) = m0 + i a0 + j a1 + k a2  (a_d = dx affine[:,d], m0 = pv - affine (f dx)):
    //   weight adjoint  Q(n) = a.x p_mass + gv(n) . mom(n)            -> folded along z, y, x into Gx,Gy,Gz
    //   M0 = sum w gv, Mx/My/Mz = sum w {i,j,k} gv                     -> gvp = M0, gaff[c][d] = dx (M_d[c] - f_d M0[c])
    //   dpos adjoint    gfx[d] = - dx (affine^T M0)[d]
    // x-planes stay a REAL loop (bounded live ranges); the plane's x-weight is selected with v_cndmask.
    R m0[3], a0[3], a1[3], a2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a0[c] = D.dx * aff[3 * c]; a1[c] = D.dx * aff[3 * c + 1]; a2[c] = D.dx * aff[3 * c + 2];
        m0[c] = pv[c] - (a0[c] * st.fx[0] + a1[c] * st.fx[1] + a2[c] * st.fx[2]);
    }
    R M0[3] = {R(0), R(0), R(0)}, Mx[3] = {R(0), R(0), R(0)}, My[3] = {R(0), R(0), R(0)}, Mz[3] = {R(0), R(0), R(0)};
    const R wz1 = st.w[1][2], wz2 = R(2) * st.w[2][2];
    // Two copies of the loop, chosen per WAVE (as in k_g2p / g2p_grad_chunk): written once with a per-lane "LDS record, overridden from global memory for a
    // drifted lane", the compiler folds the two sources into ONE flat_load through a selected generic pointer - 27 flat_load_dwordx4 per particle
    // also for the waves that never leave the tile (found in the ISA, round 3; the LDS-only copy compiles to ds_read_b128).
    auto gather = [&](auto mixed_tag) __attribute__((always_inline)) {
    constexpr bool MIXED = decltype(mixed_tag)::value;
    // (the plane's x-weight / offsets rotate through three registers: a select on i becomes an indexed read of a stack copy of the stencil here)
    R wi = st.w[0][0], wi1 = st.w[1][0], wi2 = st.w[2][0];
    int cxi = nd.cx[0], cxi1 = nd.cx[1], cxi2 = nd.cx[2];
    int txi = nd.tx[0], txi1 = nd.tx[1], txi2 = nd.tx[2];
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {
        const R fi = R(i);
        const R mi[3] = {m0[0] + fi * a0[0], m0[1] + fi * a0[1], m0[2] + fi * a0[2]};
        R s0[3] = {R(0), R(0), R(0)}, sy[3] = {R(0), R(0), R(0)}, sz[3] = {R(0), R(0), R(0)};
        R gxi = R(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const R wij = wi * st.w[j][1];
            const R mj[3] = {mi[0] + R(j) * a1[0], mi[1] + R(j) * a1[1], mi[2] + R(j) * a1[2]};
            R aq = R(0);
            R r0[3] = {R(0), R(0), R(0)}, r1[3] = {R(0), R(0), R(0)};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Vec4<R> a = gt[(!MIXED || all_in) ? txi + nd.ty[j] + nd.tz[k] : 0];      // unconditional LDS read (see k_g2p)
                if (MIXED && !all_in) a = gld(D.ain, (unsigned)(cxi + nd.cy[j] + nd.cz[k]));
                const R Q = a.x * D.p_mass + a.y * (mj[0] + R(k) * a2[0]) + a.z * (mj[1] + R(k) * a2[1]) + a.w * (mj[2] + R(k) * a2[2]);
                aq += Q * st.w[k][2];
                wg.g[k][2] += Q * wij;
                r0[0] += st.w[k][2] * a.y; r0[1] += st.w[k][2] * a.z; r0[2] += st.w[k][2] * a.w;
                if (k == 1) { r1[0] += wz1 * a.y; r1[1] += wz1 * a.z; r1[2] += wz1 * a.w; }
                if (k == 2) { r1[0] += wz2 * a.y; r1[1] += wz2 * a.z; r1[2] += wz2 * a.w; }
            }
            gxi += aq * st.w[j][1];
            wg.g[j][1] += aq * wi;
            const R wyj = st.w[j][1];
#pragma unroll
            for (int c = 0; c < 3; ++c) { s0[c] += wyj * r0[c]; sz[c] += wyj * r1[c]; sy[c] += R(j) * wyj * r0[c]; }
        }
        wg.g[0][0] += i == 0 ? gxi : R(0);
        wg.g[1][0] += i == 1 ? gxi : R(0);
        wg.g[2][0] += i == 2 ? gxi : R(0);
#pragma unroll
        for (int c = 0; c < 3; ++c) { M0[c] += wi * s0[c]; My[c] += wi * sy[c]; Mz[c] += wi * sz[c]; Mx[c] += fi * wi * s0[c]; }
        wi = wi1; wi1 = wi2; cxi = cxi1; cxi1 = cxi2; txi = txi1; txi1 = txi2;
    }
    };
    if (__all(all_in)) gather(std::false_type{});
    else gather(std::true_type{});
    SMAC_PHASE(3, KEEP);                       // 27-node gather of grid_v_in.grad
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gvp[c] = M0[c];
        gaff[3 * c + 0] = D.dx * (Mx[c] - st.fx[0] * M0[c]);
        gaff[3 * c + 1] = D.dx * (My[c] - st.fx[1] * M0[c]);
        gaff[3 * c + 2] = D.dx * (Mz[c] - st.fx[2] * M0[c]);
    }
    gfx[0] -= a0[0] * M0[0] + a0[1] * M0[1] + a0[2] * M0[2];                             // dpos = (offset - fx) dx ; a_d = dx affine[:, d]
    gfx[1] -= a1[0] * M0[0] + a1[1] * M0[1] + a1[2] * M0[2];
    gfx[2] -= a2[0] * M0[0] + a2[1] * M0[1] + a2[2] * M0[2];
    wg.to_fx(st, gfx);
    // impulse adjoint = sum_nodes w gv = gvp  -> action.grad
    if (ci >= 0)
        for (int d = 0; d < 3; ++d) atomic_add(D.action_grad + 3 * ci + d, R(6e-4) * D.dt * gvp[d]);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const R gxx = Af[rowoff(CX + d, p, D.Npad)] + D.inv_dx * gfx[d];
        Af[rowoff(CX + d, p, D.Npad)] = gxx;
        const R gvv = D.p_mass * gvp[d];
        if (ACC_VCF) Af[rowoff(CV + d, p, D.Npad)] += gvv;
        else row_store_nt(&Af[rowoff(CV + d, p, D.Npad)], gvv);
        if (KEEP) { gx_o[d] = gxx; gv_o[d] = gvv; }
    }
    // constitutive adjoint
    R gEt[9];
#if SMAC_STASH21
    R Ft[9], A1[9];                              // E and C of frame f (L2 hits): F_tmp - I is rebuilt from them here, the F_tmp adjoint below uses them again
    load_vec(Sf, CF, 9, D.Npad, p, Ft);
    load_vec(Sf, CC, 9, D.Npad, p, A1);
#endif
    {
        ConstState<CT> cs;
        CT Et[9], G[9], gFc[9], gEc[9];
#if SMAC_STASH21
        const Material<CT> mat = particle_material<CT, MAT2>(D, p);
        {
            CT Cc[9], Ec[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) { Cc[i] = (CT)A1[i]; Ec[i] = (CT)Ft[i]; }
            f_tmp(Cc, Ec, (CT)D.dt, Et);                                    // exactly what the forward half computed
        }
        cs.Jm1 = det_minus_one(Et);
#pragma unroll
        for (int i = 0; i < 9; ++i) { cs.U[i] = stash[i * BLOCK + t]; cs.V[i] = stash[(9 + i) * BLOCK + t]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            cs.e[i] = stash[(18 + i) * BLOCK + t];
            cs.ep[i] = cs.e[i];
            if (mat.ptype == MAT_PLASTIC && mat.plast == PLAST_CLIP) cs.ep[i] = min_(max_(cs.e[i], CT(-2e-3)), CT(3e-3));
        }
        if (mat.ptype == MAT_PLASTIC && mat.plast == PLAST_VON_MISES) {
            CT eh[3], nrm;
            von_mises(cs.e, mat.yield_c, cs.ep, eh, nrm);
        }
#else
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            cs.U[i] = stash[i * BLOCK + t];
            cs.V[i] = stash[(9 + i) * BLOCK + t];
            Et[i] = stash[(24 + i) * BLOCK + t];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) { cs.e[i] = stash[(18 + i) * BLOCK + t]; cs.ep[i] = stash[(21 + i) * BLOCK + t]; }
        cs.Jm1 = stash[33 * BLOCK + t];
#endif
        cs.has_svd = true;
#pragma unroll
        for (int i = 0; i < 9; ++i) { G[i] = (CT)(D.stress_scale * gaff[i]); gFc[i] = (CT)gFn[i]; }
#if !SMAC_STASH21
        const Material<CT> mat = particle_material<CT, MAT2>(D, p);
#endif
        constitutive_bwd(mat, Et, cs, G, gFc, gEc);
#pragma unroll
        for (int i = 0; i < 9; ++i) gEt[i] = (R)gEc[i];
    }
    SMAC_PHASE(4, KEEP);                       // constitutive adjoint
    // compute_F_tmp.grad: F_tmp = (I + dt C)(I + E)
    R gC[9], gE[9];
#if !SMAC_STASH21
    R Ft[9], A1[9];
    load_vec(Sf, CF, 9, D.Npad, p, Ft);          // (C, E come back from L2: cheaper than 18 more stash slots)
    load_vec(Sf, CC, 9, D.Npad, p, A1);
#endif
#pragma unroll
    for (int i = 0; i < 9; ++i) A1[i] *= D.dt;
    Ft[0] += R(1); Ft[4] += R(1); Ft[8] += R(1);
    A1[0] += R(1); A1[4] += R(1); A1[8] += R(1);
    mmt(gEt, Ft, gC);          // gEt (I+E)^T
    mtm(A1, gEt, gE);          // (I + dt C)^T gEt
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const R gc = D.dt * gC[i] + D.p_mass * gaff[i];
        if (ACC_VCF) {
            Af[rowoff(CC + i, p, D.Npad)] += gc;
            Af[rowoff(CF + i, p, D.Npad)] += gE[i];
        } else {
            row_store_nt(&Af[rowoff(CC + i, p, D.Npad)], gc);
            row_store_nt(&Af[rowoff(CF + i, p, D.Npad)], gE[i]);
        }
        if (KEEP) gC_o[i] = gc;
    }
    SMAC_PHASE(5, KEEP);                       // F_tmp adjoint, 18 rows stored
}

template <class R, bool ACC_VCF, bool PCON, bool MAT2 = false>
__global__ __launch_bounds__(BLOCK, occ<R>::heavy) void k_p2g_grad(DevSim<R> D, int f) {
    typedef typename const_t<R>::type CT;
    __shared__ CT stash[STASH * BLOCK];
    __shared__ Vec4<R> gt[PTILE];
    SMAC_CHUNK_PROLOGUE
    gather_tile_load_p(D, D.ain, ch.block, gt);
    __syncthreads();
    if (!valid) return;
    p2g_grad_particle<R, ACC_VCF, PCON, false, MAT2>(D, f, ch, p, t, stash, gt, (R*)nullptr, (R*)nullptr, (R*)nullptr, D.An_map ? D.An_map[p] : p);
}

// k_p2g_grad of substep f and k_g2p_grad of substep f - 1 in one launch.  Between two re-sorts a particle keeps its chunk, and the adjoint of
// (x, v, C) at frame f that p2g.grad completes is exactly the input of the G2P adjoint one substep earlier: it stays in registers (15 rows of
// reads, a kernel boundary and a workgroup prologue less per backward substep; the rows are still written, so every adjoint frame stays readable).
// The host restores the forward grid of substep f - 1 BEFORE this launch (k_grid_restore leaves grid_v_in.grad alone), so `grid_v_out` is the
// earlier substep's while `grid_v_in.grad` is still this one's.  D.Af_prev: adjoint frame f - 1.  ACC_X: frame f - 1 carries a seed.
template <class R, bool ACC_X>
__global__ __launch_bounds__(BLOCK, (sizeof(R) == 4 ? occ<R>::heavy : 1)) void k_p2g_g2p_grad(DevSim<R> D, int f) {      // (f64: never launched, see can_fuse_prev)
    typedef typename const_t<R>::type CT;
    typedef typename ScatterTile<R>::word W;
    __shared__ CT stash[STASH * BLOCK];
    // [scatter tile of grid_v_out.grad of substep f - 1, in words W | gather tile of grid_v_in.grad, grid_m.grad of substep f].  A sparse chunk's f64 scatter tile
    // does not fit the first part (wide tiles, 3 workgroups per CU): it lies over BOTH, and is zeroed once the gather tile is dead (one more barrier for those chunks)
    constexpr int SCATTER_BYTES = 3 * PTILE * (int)sizeof(W);
    constexpr bool OVERLAY = sizeof(R) == 4 && SCATTER_BYTES < 3 * PTILE * (int)sizeof(double);
    static_assert(!OVERLAY || SCATTER_BYTES + PTILE * (int)sizeof(Vec4<R>) >= 3 * PTILE * (int)sizeof(double), "sparse scatter tile must fit over scatter + gather tile");
    __shared__ __attribute__((aligned(16))) unsigned char pool[SCATTER_BYTES + PTILE * sizeof(Vec4<R>)];
    __shared__ Vec4<R> gtv[PTILE];              // grid_v_out of substep f - 1
    __shared__ R smax[4];
    SMAC_PHASE(10, true);
    SMAC_CHUNK_PROLOGUE
    SMAC_PHASE(11, ch.count >= 0);
    W* const tile = (W*)pool;
    double* const tile64 = (double*)pool;
    double* const tile_raw = (double*)pool;
    Vec4<R>* const gt = (Vec4<R>*)(pool + SCATTER_BYTES);
    const bool sparse = sizeof(R) == 4 && ch.count <= SPARSE_MAX;
    SMAC_PHASE(0, valid);
    if (!sparse) lds_zero16(pool, SCATTER_BYTES);
    else if (!OVERLAY) lds_zero16(pool, 3 * PTILE * (int)sizeof(double));
    gather_tile_load_p(D, D.ain, ch.block, gt);
    gather_tile_load_p(D, D.vout, ch.block, gtv);
    typename pos_of<R>::type xp[3] = {pos_mid<R>(), pos_mid<R>(), pos_mid<R>()};
    if (valid) load_pos(frame(D.S, f - 1, D.Npad), D.Npad, p, xp);
    __syncthreads();
    SMAC_PHASE(1, valid);                      // both gather tiles staged
    R gx[3] = {R(0), R(0), R(0)}, gnv[3] = {R(0), R(0), R(0)}, gC1[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) gC1[c] = R(0);
    if (valid) {
        R gv[3];
        p2g_grad_particle<R, false, true, true>(D, f, ch, p, t, stash, gt, gx, gv, gC1, D.An_map ? D.An_map[p] : p);
        const R four_inv_dx = R(4) * D.inv_dx;
#pragma unroll
        for (int c = 0; c < 3; ++c) gnv[c] = gv[c] + D.dt * gx[c];                           // x' = x + dt v'
#pragma unroll
        for (int c = 0; c < 9; ++c) gC1[c] *= four_inv_dx;
    }
    if (OVERLAY && sparse) {                     // (workgroup-uniform) every wave is done with the gather tile: its space becomes part of the f64 scatter tile
        __syncthreads();
        lds_zero16(pool, 3 * PTILE * (int)sizeof(double));               // (made visible by the barrier inside tile_scale)
    }
    g2p_grad_chunk<R, ACC_X>(D, ch, p, t, valid, xp, gx, gnv, gC1, D.Af_prev, tile_raw, gtv, smax);
}

// ------------------------------------------------------------------------------------------
// halo planes for the slab decomposition: x-planes [plane0, plane0+np) of one 4-scalar field,
// packed dense (np, n, n) for RCCL send/recv.  Cells of blocks that are not active in this epoch read as
// zero / are not written (their storage may hold stale data from another epoch).
// ------------------------------------------------------------------------------------------
template <class R>
__global__ __launch_bounds__(BLOCK) void k_halo_pack(DevSim<R> D, const Vec4<R>* field, const Vec4<R>* minus, int plane0, int np, Vec4<R>* out) {
    const int idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= np * D.n * D.n) return;
    const int k = idx % D.n, j = (idx / D.n) % D.n, i = plane0 + idx / (D.n * D.n);
    Vec4<R> v = {R(0), R(0), R(0), R(0)};
    if (D.block_active[block_of(D.nb, i, j, k)]) {
        const size_t c = cell_of(D.nb, i, j, k);
        v = field[c];
        if (minus) { const Vec4<R> m = minus[c]; v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w; }
    }
    out[idx] = v;
}
template <class R>
__global__ __launch_bounds__(BLOCK) void k_halo_unpack_add(DevSim<R> D, Vec4<R>* field, int plane0, int np, const Vec4<R>* in) {
    const int idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= np * D.n * D.n) return;
    const int k = idx % D.n, j = (idx / D.n) % D.n, i = plane0 + idx / (D.n * D.n);
    if (!D.block_active[block_of(D.nb, i, j, k)]) return;
    const size_t c = cell_of(D.nb, i, j, k);
    const Vec4<R> a = in[idx];
    Vec4<R> v = field[c];
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    field[c] = v;
}

// both neighbours' planes in ONE launch (the in-library exchange, softmac_hip.hip `exchange`): blockIdx.y picks the entry of `hs`;
// buffer slot 0 = left neighbour, 1 = right neighbour, each np * n * n records
template <class R>
__global__ __launch_bounds__(BLOCK) void k_halo_pack2(DevSim<R> D, const Vec4<R>* field, const Vec4<R>* minus, HaloSides hs, int np, Vec4<R>* out) {
    const int idx = blockIdx.x * BLOCK + threadIdx.x, total = np * D.n * D.n;
    if (idx >= total) return;
    const int s = blockIdx.y;
    const int k = idx % D.n, j = (idx / D.n) % D.n, i = hs.plane0[s] + idx / (D.n * D.n);
    Vec4<R> v = {R(0), R(0), R(0), R(0)};
    if (D.block_active[block_of(D.nb, i, j, k)]) {
        const size_t c = cell_of(D.nb, i, j, k);
        v = field[c];
        if (minus) { const Vec4<R> m = minus[c]; v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w; }
    }
    out[(size_t)hs.slot[s] * total + idx] = v;
}
template <class R>
__global__ __launch_bounds__(BLOCK) void k_halo_unpack_add2(DevSim<R> D, Vec4<R>* field, HaloSides hs, int np, const Vec4<R>* in) {
    const int idx = blockIdx.x * BLOCK + threadIdx.x, total = np * D.n * D.n;
    if (idx >= total) return;
    const int s = blockIdx.y;
    const int k = idx % D.n, j = (idx / D.n) % D.n, i = hs.plane0[s] + idx / (D.n * D.n);
    if (!D.block_active[block_of(D.nb, i, j, k)]) return;
    const size_t c = cell_of(D.nb, i, j, k);
    const Vec4<R> a = in[(size_t)hs.slot[s] * total + idx];
    Vec4<R> v = field[c];
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    field[c] = v;
}

// ------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------
// compute_grid_m_kernel :607-617 - dense (i,j,k) row-major output, particle order irrelevant
template <class R>
__global__ void k_grid_m_only(const R* x0, const R* x1, const R* x2, int N, int n, R inv_dx, R p_mass, R* out) {
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= N) return;
    typedef typename pos_of<R>::type PX;
    const size_t po = poff(p);
    const PX x[3] = {((const PX*)x0)[po], ((const PX*)x1)[po], ((const PX*)x2)[po]};
    Stencil<R> st;
    make_stencil_pos(x, n, st);
    (void)inv_dx;
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
// `stride` scalars separate the primitives' state arrays; thread i advances primitive i (one launch for all of them, :329-331)
template <class R>
__global__ void k_prim_fk(R* state, int f, R dt, int nprims = 1, size_t stride = 0) {      // instantiated with R = double: primitive state is f64 in both modes
    if ((int)threadIdx.x >= nprims || blockIdx.x != 0) return;
    prim_fk_step(state + threadIdx.x * stride, f, dt);
}
template <class R>
__global__ void k_prim_fk_grad(const R* state, R* grad, int f, R dt, size_t stride = 0) {   // one workgroup per primitive (they do not interact: :367-369 in any order)
    const int dir = threadIdx.x;
    if (dir >= 13) return;
    prim_fk_step_grad(state + blockIdx.x * stride, grad + blockIdx.x * stride, f, dt, dir);
}


// Velocity control on the device (primitive_base.py:285-319, rigid_simulator_vel.py:20-44): set_action writes the 6-D action (w, v) into the
// action buffer and fans it out over the env step's frames; get_action_grad folds the frames' v / w adjoints back into action_buffer.grad.
struct Act6 { double a[6]; };
__global__ void k_prim_set_action(double* state, double* action_buf, int s, int n, Act6 a) {
    const int j = threadIdx.x;
    if (j < 6) action_buf[(size_t)s * 6 + j] = a.a[j];
    if (j < n) {
        double* st = state + (size_t)(s * n + j) * 13;
        for (int k = 0; k < 3; ++k) { st[7 + k] = a.a[3 + k]; st[10 + k] = a.a[k]; }
    }
}
__global__ void k_prim_action_grad(const double* grad, double* action_buf_grad, int s, int n) {      // one workgroup per env step: s + blockIdx.x
    const int c = threadIdx.x;
    if (c >= 6) return;
    s += (int)blockIdx.x;
    double acc = action_buf_grad[(size_t)s * 6 + c];
    for (int j = 0; j < n; ++j) acc += grad[(size_t)(s * n + j) * 13 + (c < 3 ? 10 + c : 7 + (c - 3))];     // set_velocity_from_action_kernel.grad :315-319
    action_buf_grad[(size_t)s * 6 + c] = acc;
}

}  // namespace smac
