// libsoftmac_hip.so - C ABI (include/softmac_hip.h) over the HIP kernels in smac_kernels.hpp.
// Host side only: handle, device memory, launch sequencing, f64 <-> device-layout conversion.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/softmac_hip.h"
#include "smac_kernels.hpp"
#include "smac_comm.hpp"
#include "smac_migrate.hpp"
#include "smac_cloth_kernels.hpp"
#include "smac_voxel.hpp"
#include "smac_loss.hpp"

using namespace smac;

static thread_local std::string g_create_error;

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            char _b[512];                                                                     \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            this->err = _b;                                                                   \
            return SMAC_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define REQUIRE(cond, msg)                   \
    do {                                     \
        if (!(cond)) {                       \
            this->err = std::string(msg);    \
            return SMAC_ERR_INVALID;         \
        }                                    \
    } while (0)

enum KernelId { K_CLEAR = 0, K_CKPT, K_P2G, K_GRID_OP, K_CONTACT, K_G2P, K_G2P_GRAD, K_REDUCE, K_CONTACT_GRAD, K_GRID_OP_GRAD, K_P2G_GRAD, K_FK, K_SORT, K_REORDER, K_P2G_G2P_GRAD, K_G2P_P2G, K_COUNT };
static const char* kDriftMessage =
    "a particle left the halo of its grid block between two re-sorts (it more than doubled its speed inside one re-sort "
    "interval): the frames after that substep are invalid - lower sort_interval or dt";
static const char* kSlabLeftMessage =
    "slab decomposition: a particle's stencil left the x-planes this rank shares with its neighbours (its deposits there would be lost): "
    "migrate more often (SlabRunner.migrate at every re-sort) or widen the shared band (nplanes = 2 + 2 * drift tolerance)";
static const char* kKernelNames[K_COUNT] = {"clear_grid", "grid_checkpoint", "p2g", "grid_op", "contact", "g2p", "g2p_grad", "reduce_agvout", "contact_grad",
                                            "grid_op_grad", "p2g_grad", "forward_kinematics", "sort", "reorder_adjoint", "p2g_g2p_grad", "g2p_p2g"};

// a few scalars handed to the device inside the kernel's argument block
template <class T, int N> struct SmallArgs { T v[N]; };
template <class T, int N>
__global__ void k_set_small(SmallArgs<T, N> a, int n, T* dst, T* zero_too) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        dst[i] = a.v[i];
        if (zero_too) zero_too[i] = T(0);
    }
}

struct ISim {
    std::string err;
    virtual ~ISim() {}
    virtual int init(const smac_config& c) = 0;
    virtual int sync() = 0;
    virtual int reset(const double* state, int cols) = 0;
    virtual int set_frame(int f, const double* x, const double* v, const double* F, const double* C) = 0;
    virtual int get_frame(int f, double* x, double* v, double* F, double* C) = 0;
    virtual int get_state24(int f, double* out) = 0;
    virtual int copy_frame(int src, int dst) = 0;
    virtual int get_grad(int f, double* gx, double* gv, double* gF, double* gC) = 0;
    virtual int add_grad(int f, const double* gx, const double* gv, const double* gF, const double* gC) = 0;
    virtual int add_grad_device(int f, const double* gx, const double* gv, const double* gF, const double* gC) = 0;
    virtual int clear_grads() = 0;
    virtual int carry_grad(int src, int dst) = 0;
    virtual int set_control_idx(const int32_t* idx) = 0;
    virtual int set_material_ids(const int32_t* ids) = 0;
    virtual int set_action_v(const double* action) = 0;
    virtual int get_action_grad(double* out) = 0;
    virtual int set_segment(int n_live, int frame_shift) = 0;
    virtual int compute_grid_m(int f, double* out) = 0;
    virtual int substep(int f, const double* action) = 0;
    virtual int substep_grad(int f, const double* action, const double* ext_f_grad, double* action_grad_out) = 0;
    virtual int prim_upload_sdf(int prim, const double* sdf, const double* normal, const int32_t* res, const double* lower,
                                const double* upper, double dx) = 0;
    virtual int prim_set_params(int prim, double friction, double softness, int contact) = 0;
    virtual int prim_set_state(int prim, int f0, int f1, const double* s13) = 0;
    virtual int prim_get_state(int prim, int f, double* s13) = 0;
    virtual int prim_get_state_grad(int prim, int f0, int f1, double* g13) = 0;
    virtual int prim_set_states(int prim, int f0, int f1, const double* s13) = 0;
    virtual int prim_get_state_grads(int prim, int f0, int f1, double* g13) = 0;
    virtual int prim_add_state_grad(int prim, int f, const double* g13) = 0;
    virtual int prim_fk(int prim, int f) = 0;
    virtual int prim_fk_grad(int prim, int f) = 0;
    virtual int prim_get_ext_f(int prim, double* e6) = 0;
    virtual int prim_clear_ext_f(int prim) = 0;
    virtual int prim_set_action(int prim, int s, int n, const double* a6) = 0;
    virtual int prim_get_action_grad(int prim, int s, int n, double* g6) = 0;
    virtual int prim_get_action_grads(int prim, int s0, int s1, int n, double* g6) = 0;
    virtual int prim_reset(int prim) = 0;
    virtual int timer_start() = 0;
    virtual int timer_stop(double* ms) = 0;
    virtual int profile_enable(int on) = 0;
    virtual int profile_reset() = 0;
    virtual int profile_get(int i, char* name, int cap, double* ms, int64_t* launches) = 0;
    virtual int count_active_cells(int f, int64_t* cells) = 0;
    virtual int contact_counts(int32_t* nhits, int32_t* nchunks_hit) = 0;
    virtual int loss_set_target(const double* target, int m) = 0;
    virtual int loss_chamfer(int f, double weight, int add_grad, double* loss_out) = 0;
    virtual int loss_min_dist(int f, int id0, int id1, const double* c3, double offset, double weight, int add_grad, double* out4) = 0;
    virtual int grid_ptr(const char* field, void** p, int64_t* n, int32_t* bytes) = 0;
    virtual int substep_phase_v(int f, int phase) = 0;
    virtual int substep_grad_phase_v(int f, const double* ext_f_grad, int phase) = 0;
    virtual int halo_pack(const char* field, int plane0, int np, void* dev_out, int minus_mixed) = 0;
    virtual int halo_unpack_add(const char* field, int plane0, int np, const void* dev_in) = 0;
    virtual int set_stream(void* s) = 0;
    virtual int comm_init(const char* id128, int rank, int world) = 0;
    virtual int comm_slab(int left0, int right0, int nplanes, int contact_left, int contact_right, int base_lo, int base_hi, int self_loop) = 0;
    virtual int substeps_slab(int f0, int count) = 0;
    virtual int substeps_slab_grad(int f0, int count, const double* ext_f_grad) = 0;
    virtual int comm_allreduce_ext_f(double* total_out, int clear) = 0;
    virtual int comm_allreduce_prim_grad(int f0, int f1) = 0;
    virtual int comm_destroy() = 0;
    virtual int comm_abort() = 0;
    virtual int migrate(int f, int base_lo, int base_hi, int32_t* out3) = 0;
    virtual int migrate_grad() = 0;
    virtual int set_ids(const int64_t* ids) = 0;
    virtual int get_ids(int64_t* ids) = 0;
    virtual int stream_handle(void** s) = 0;
    virtual void hint_backward_next(int f) = 0;
    virtual void hint_forward_next(int f) = 0;
    virtual int set_param(const char* name, double value) = 0;
    virtual int get_param(const char* name, double* value) = 0;
    virtual int cloth_create(int nv, int nf, const int32_t* faces, int nn, const int32_t* nbr, const int8_t* nbr_dir, double friction,
                             double softness, double force_scale, int sticky, double scale) = 0;
    virtual int cloth_set_state(int f0, int f1, const double* pos, const double* vel) = 0;
    virtual int cloth_get_state(int f, double* pos, double* vel, int grad) = 0;
    virtual int cloth_ext_f(int op, double* buf) = 0;
    virtual int cloth_contact(int op, int f) = 0;
    virtual int cloth_get_contact(int f, int32_t* ids, int8_t* pen) = 0;
    virtual int cloth_set_contact(int f, const int32_t* ids, const int8_t* pen) = 0;
    virtual int cloth_check_penetration(int f, int32_t* total, int32_t* warnings) = 0;
};

template <class R> struct Sim final : ISim {
    smac_config cfg{};
    DevSim<R> D{};
    hipStream_t stream = nullptr;
    Vec4<R>* grid_block = nullptr;   // 6 fields of G 4-scalar records: vin, vmix, vout, then their adjoints
    R* prim_tables[SMAC_MAX_PRIMS][2] = {};
    double* prim_tables64[SMAC_MAX_PRIMS][2] = {};   // f64 copies for the forecast contact chain (the same buffers when R = double)
    double* action_buf = nullptr;      // per-primitive velocity-control action buffer [P][max_frames][6] + its grad
    double* action_buf_grad = nullptr;
    unsigned long long* d_counter = nullptr;
    int* d_control_idx = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool profiling = false;
    struct Rec { int id; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    double prof_ms[K_COUNT] = {};
    int64_t prof_n[K_COUNT] = {};

    // ---- epochs: one per re-sort.  Epoch 0 is the identity order of user-provided frames.
    struct Epoch {
        int* orig = nullptr;        // sorted slot -> original id (device)
        int* from_prev = nullptr;   // slot in the epoch this one was sorted FROM -> slot here (the re-sort's destination map): an adjoint frame of this
        int prev = -1;              // epoch goes back to that order by one gather through it, without the two inverse / compose passes
        long long serial = 0, prev_serial = -1;   // (epoch slots are recycled: `prev` names the same epoch only while the serials match)
        int* cellrank = nullptr;    // sorted slot -> (cell << 5 | rank in the cell) at this binning: the next re-sort keeps the ranks of particles that stayed (smac_sort.hpp)
        int* inv = nullptr;         // original id -> sorted slot (device, lazy)
        bool inv_valid = false;
        Chunk* chunks = nullptr;
        int nchunks = 0;
        int* active = nullptr;
        int nactive = 0;
        int* block_chunk_start = nullptr;
        int* block_chunks = nullptr;
        int* block_active = nullptr;
        int* block_slot = nullptr;    // dense per block: slot in the active list (exclusive scan of the active flags)
        int* tail_expect = nullptr;   // dense per block: chunks of the 27 blocks around it (tail reduction's arrival count)
        int frame = 0;              // frame at which the sort happened
        int interval = 1;           // substeps this binning is used for (<= sort_interval, shortened for fast particles)
        bool live = false;
    };
    std::vector<Epoch> epochs;      // [0] = identity
    std::vector<Epoch> epoch_pool;  // buffer sets of dropped epochs, reused by the next re-sort (all sizes are worst-case,
                                    // so a re-sort neither allocates nor frees - hipFree would drain the stream)
    std::vector<int> frame_epoch;   // order tag of S[f]   (-1: never written)
    std::vector<int> adj_epoch;     // order tag of A[f]   (-1: all zero, any order)
    std::vector<char> adj_stale;    // A[f] is logically zero (adj_epoch -1) but its memory has not been cleared yet
    int grid_epoch = 0;             // epoch whose active blocks may hold non-zero grid data
    int sort_interval = 8;
    int nblocks = 0;
    // sort scratch
    unsigned long long* d_bin_mask = nullptr;
    int* d_over_prefix = nullptr;
    unsigned* d_vmax = nullptr;
    float* d_vmax_part = nullptr;
    int *d_cell_count = nullptr, *d_bin = nullptr, *d_bin_start = nullptr, *d_key = nullptr, *d_slot = nullptr, *d_dest = nullptr;
    int *d_block_start = nullptr, *d_block_chunks = nullptr, *d_chunk_start = nullptr, *d_active_flag = nullptr, *d_active_start = nullptr;
    int* d_guest = nullptr;          // 4 x (nblocks + 1): the blocks' particle counts, room reserved in them, direction and tickets of the blocks that hand particles over (smac_sort.hpp "guests")
    int* d_map = nullptr;
    void* d_cub = nullptr;
    size_t cub_bytes = 0;
    R* tmp_frame = nullptr;         // NCOMP*Npad scratch (sort moves, re-ordered adjoints)
    R* tmp_frame2 = nullptr;        // second one, allocated when the adjoints of frames f AND f+1 both arrive in another particle order
    Vec4<R>* slab = nullptr;
    size_t slab_chunks = 0;
    int* h_sort_info = nullptr;      // pinned: what sort_frame reads back of a re-sort (k_sort_info)
    int* last_dest = nullptr;        // the destination map of the latest re-sort (statistics)
    Hit* d_hits = nullptr;           // capacity Npad
    Hit* d_hits2 = nullptr;          // the hit list of odd frames (k_g2p_p2g appends the next substep's hits while this substep's list is filed)
    Vec4<R>* vdrift = nullptr;       // DevSim::vdrift
    int* d_tail_cnt = nullptr;       // DevSim::tail_cnt
    // (only in builds that carry it, -DSMAC_TAIL_BUILD=1: it measured slower than the launches it replaces and the shipped kernels are compiled without it)
    int tail_env = SMAC_TAIL_BUILD ? (getenv("SMAC_TAIL_REDUCE") ? atoi(getenv("SMAC_TAIL_REDUCE")) : 1) : 0;
    int p2g_tail_frame = -1;         // substep whose P2G ended with the tail reduction: {m,p} and v_out are complete, no k_grid_op launch
    int tail_bwd_env = getenv("SMAC_TAIL_REDUCE_BWD") ? atoi(getenv("SMAC_TAIL_REDUCE_BWD")) : 1;    // the backward half of it by itself (A/B)
    int bwd_tail_frame = -1;         // substep whose grid adjoint pass (reduction + grid_op's node adjoint) ran inside the fused launch of the substep after it
    bool pend_restore = false, pend_fk = false;   // ... and what the reduction launch used to carry: restore of the next frame's grid, forward_kinematics.grad
    int* d_nhits = nullptr;          // [0] = nhits, [1] = ncand
    int* d_cand = nullptr;
    int* d_pmask = nullptr;
    int* d_drift = nullptr;
    R* dense_tmp = nullptr;
    // grid checkpoints (one slot per frame in one arena); see k_grid_save
    Vec4<R>* ck_arena = nullptr;
    Hit* ck_hits = nullptr;          // per frame: the contact hit list of that substep (capacity ck_hit_cap), with its length
    int ck_hit_cap = 0;
    int* ck_nhits = nullptr;
    // pinned, device-visible: the hit count filed with each frame's checkpoint, written by the saving launch (-1: unknown).  substep_grad reads it after
    // the synchronisation every backward pass starts with: a frame whose list is empty needs no contact adjoint launch (20 us of fixed latency).
    int* h_nhits = nullptr;
    int hits_from_ck_frame = -1;         // frame whose hit list the current substep_grad call took from the checkpoint (copied or in place)
    int contact_skips = 0;               // contact adjoint launches saved that way (smac_get_param "contact_skips")
    unsigned char* ck_empty = nullptr;   // [frame][active slot]: the block held no mass and nothing was filed for it (DevSim::ck_flags)
    int ck_skip_empty = getenv("SMAC_CK_SKIP_EMPTY") ? atoi(getenv("SMAC_CK_SKIP_EMPTY")) : 1;
    // the flags of frame f for the backward grid pass: only while the frame's checkpoint (and with it the flags) is the one this epoch filed
    unsigned char* reduce_flags(int f, int e) { return (ck_arena && ck_epoch[f] == e && ck_gen[f] == config_gen) ? ck_flags_of(f) : (unsigned char*)nullptr; }
    unsigned char* ck_flags_of(int f) { return (ck_empty && ck_skip_empty) ? ck_empty + (size_t)f * ck_slot_blocks : (unsigned char*)nullptr; }
    size_t ck_slot_blocks = 0;       // capacity of a slot in grid blocks
    bool ck_enabled = true, ck_tried = false;
    // (The forward grid reaches the checkpoint through copy work riding in k_g2p's launch and comes back through the restore riding in the grid-adjoint
    // reduction's.  Writing / reading the checkpoint in place was measured in round 2 and was slower or equal: profiles/r02_x_checkpoint_in_place.txt.)
    std::vector<int> ck_epoch;       // epoch the slot of frame f was saved in (-1: invalid)
    std::vector<char> ck_has_hits;   // the frame's contact hit list is on file too
    std::vector<long long> ck_gen;   // configuration generation at save time
    long long config_gen = 0;        // bumped whenever something the forward grid depends on may have changed

    ~Sim() override {
        if (stream) hipStreamSynchronize(stream);
#if SMAC_PHASE_CLOCK
        if (const char* path = getenv("SMAC_PHASE_DUMP")) {          // tools/phase_clock.sh: per marker, sum of timestamps (mod 2^64) and hits
            static unsigned long long hs[48 * 64], hc[48 * 64];
            (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(smac::smac_phase_sum), sizeof(hs));
            (void)hipMemcpyFromSymbol(hc, HIP_SYMBOL(smac::smac_phase_cnt), sizeof(hc));
            if (FILE* fp = fopen(path, "a")) {
                for (int m = 0; m < 48; ++m) {
                    unsigned long long a = 0, c = 0;
                    for (int k = 0; k < 64; ++k) { a += hs[m * 64 + k]; c += hc[m * 64 + k]; }
                    fprintf(fp, "%d %llu %llu\n", m, a, c);
                }
                static unsigned long long hh[2 * 64];                  // per-wave durations of the two contact kernels: "-1 kernel*64+bin count"
                (void)hipMemcpyFromSymbol(hh, HIP_SYMBOL(smac::smac_phase_hist), sizeof(hh));
                for (int k = 0; k < 2 * 64; ++k)
                    if (hh[k]) fprintf(fp, "-1 %d %llu\n", k, hh[k]);
                static unsigned long long slow[256 * 8];               // the slowest waves of k_contact_grad: "-2 record*8+word value"
                unsigned ns = 0;
                (void)hipMemcpyFromSymbol(slow, HIP_SYMBOL(smac::smac_slow), sizeof(slow));
                (void)hipMemcpyFromSymbol(&ns, HIP_SYMBOL(smac::smac_slow_n), sizeof(ns));
                for (unsigned k = 0; k < (ns < 256u ? ns : 256u) * 8; ++k) fprintf(fp, "-2 %u %llu\n", k, slow[k]);
                fclose(fp);
            }
        }
#endif
        comm_destroy();
        hipFree(D.S); hipFree(D.A); hipFree(grid_block); hipFree(grid_alt); hipFree(D.prim_state); hipFree(D.prim_grad);
        hipFree(D.ext_f); hipFree(D.action); hipFree(D.action_grad); hipFree(d_control_idx); hipFree(d_counter);
        hipFree(action_buf); hipFree(action_buf_grad); hipFree(scratch);
        for (auto& e : epochs) free_epoch(e);
        for (auto& e : epoch_pool) free_epoch(e);
        hipFree(d_bin_mask); hipFree(d_over_prefix); hipFree(d_vmax); hipFree(d_vmax_part);
        hipFree(d_cell_count); hipFree(d_bin); hipFree(d_bin_start); hipFree(d_key); hipFree(d_slot); hipFree(d_dest);
        hipFree(d_block_start); hipFree(d_block_chunks); hipFree(d_chunk_start); hipFree(d_active_flag); hipFree(d_active_start); hipFree(d_guest);
        hipFree(d_map); hipFree(d_map_an); hipFree(d_cub); hipFree(tmp_frame); hipFree(tmp_frame2); hipFree(slab); hipFree(d_drift); hipFree(dense_tmp);
        for (PIdx* I : {&pi_target, &pi_cur}) { hipFree(I->cell_start); hipFree(I->count); hipFree(I->key); hipFree(I->ids); hipFree(I->slots); hipFree(I->pts); }
        hipFree(d_io); hipFree(ext_snap); hipFree(cloth_ext_snap);
        for (auto& m : migs) { hipFree(m.src_slot); hipFree(m.ids_old); }
        hipFree(d_ids); hipFree(mig_flags); hipFree(mig_pos); hipFree(mig_counts);
        for (char* b : mig_buf) hipFree(b);
        hipFree(fhash.count); hipFree(fhash.start); hipFree(fhash.list); hipFree(fhash.overflow);
        if (hash_total_host) hipHostFree(hash_total_host);
        hipFree(d_cloth_faces); hipFree(d_cloth_nbr); hipFree(d_cloth_nbr_dir); hipFree(d_cloth_warn); hipFree(d_cloth_ext_scratch);
        hipFree(D.cloth.pos); hipFree(D.cloth.vel); hipFree(D.cloth.pos_grad); hipFree(D.cloth.vel_grad); hipFree(D.cloth.ext_f);
        hipFree(D.cloth.ext_f_grad); hipFree(D.cloth.contact_id); hipFree(D.cloth.penetration); hipFree(D.cloth.contact_before);
        hipFree(d_target); hipFree(d_loss); hipFree(d_best); hipFree(d_md_out);
        hipFree(d_mat_id);
        hipFree(d_hits); hipFree(d_hits2); hipFree(vdrift); hipFree(d_tail_cnt); hipFree(d_nhits); hipFree(d_cand); hipFree(d_pmask); hipFree(ck_arena); hipFree(ck_hits); hipFree(ck_nhits); hipFree(ck_empty);
        if (h_nhits) hipHostFree(h_nhits);
        if (h_sort_info) hipHostFree(h_sort_info);
        for (int i = 0; i < SMAC_MAX_PRIMS; ++i) {
            if ((void*)prim_tables64[i][0] != (void*)prim_tables[i][0]) { hipFree(prim_tables64[i][0]); hipFree(prim_tables64[i][1]); }
            hipFree(prim_tables[i][0]); hipFree(prim_tables[i][1]);
        }
        for (auto e : pool) hipEventDestroy(e);
        for (auto& r : recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
        if (t0) hipEventDestroy(t0);
        if (t1) hipEventDestroy(t1);
        if (stream && own_stream) hipStreamDestroy(stream);
    }

    size_t frame_scalars() const { return (size_t)NCOMP * D.Npad; }
    int nblk(size_t n) const { return (int)((n + BLOCK - 1) / BLOCK); }

    int init(const smac_config& c) override {
        cfg = c;
        REQUIRE(c.n_particles > 0 && c.n_grid >= 8 && c.max_frames >= 2, "bad sizes");
        REQUIRE(c.n_primitives >= 0 && c.n_primitives <= SMAC_MAX_PRIMS, "n_primitives > SMAC_MAX_PRIMS");
        REQUIRE(c.substeps >= 1, "substeps < 1");
        REQUIRE(c.n_grid % 4 == 0, "n_grid must be a multiple of 4 (4x4x4 grid blocks)");
        HIP_TRY(hipSetDevice(c.device));
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&t0));
        HIP_TRY(hipEventCreate(&t1));
        D.N = c.n_particles;
        D.Npad = (c.n_particles + 255) / 256 * 256;
        // component rows are Npad scalars apart: keep that stride off large powers of two (1M particles would put all
        // 24 rows of a particle on the same HBM channel phase).  SMAC_ROW_SKEW scalars, a multiple of 64 (256 B).
        if (D.Npad >= 65536) D.Npad += 4160;            // (65 x 64 scalars; profiles/r02_f_frame_layout.txt)
        if (SMAC_TILE_P) D.Npad = (c.n_particles + SMAC_TILE_P - 1) / SMAC_TILE_P * SMAC_TILE_P;       // AoSoA tiles (smac_math.hpp "Frame layout")
        D.n = c.n_grid;
        D.G = (size_t)c.n_grid * c.n_grid * c.n_grid;
        D.P = c.n_primitives;
        D.n_control = c.n_control;
        D.substeps = c.substeps;
        D.collision_type = c.collision_type;
        D.sticky = c.ground_friction >= 10.0 ? 1 : 0;
        D.max_frames = c.max_frames;
        D.dt = (R)c.dt;
        D.dt64 = c.dt;
        D.inv_dx = (R)c.n_grid;
        D.dx = (R)(1.0 / c.n_grid);
        D.p_mass = (R)c.p_mass;
        D.stress_scale = (R)(-c.dt * c.p_vol * 4.0 * (double)c.n_grid * (double)c.n_grid);    // mpm_simulator.py:247
        for (int d = 0; d < 3; ++d) D.g[d] = (R)c.gravity[d];
        D.mat.ptype = c.ptype; D.mat.model = c.material_model; D.mat.mu = (R)c.mu; D.mat.lam = (R)c.lam;
        D.mat.plast = PLAST_CLIP; D.mat.yield_c = R(0); D.m_eps = R(1e-10);
        const size_t fs = frame_scalars() * sizeof(R);
        HIP_TRY(hipMalloc((void**)&D.S, fs * c.max_frames));
        HIP_TRY(hipMemsetAsync(D.S, 0, fs * c.max_frames, stream));
        if (c.grad_enabled) {
            adj_slots = c.max_frames;
            if (c.adjoint_frames > 0 && c.adjoint_frames < c.max_frames) adj_slots = c.adjoint_frames < 3 ? 3 : c.adjoint_frames;
            HIP_TRY(hipMalloc((void**)&D.A, fs * adj_slots));
            HIP_TRY(hipMemsetAsync(D.A, 0, fs * adj_slots, stream));
            adj_slot.assign(c.max_frames, -1);
            adj_evicted.assign(c.max_frames, 0);
            free_slots.clear();
            for (int i = adj_slots - 1; i >= 0; --i) free_slots.push_back(i);
        }
        HIP_TRY(hipMalloc((void**)&grid_block, 6 * D.G * sizeof(Vec4<R>)));
        HIP_TRY(hipMemsetAsync(grid_block, 0, 6 * D.G * sizeof(Vec4<R>), stream));
        D.vin = grid_block; D.vmix = grid_block + D.G; D.vout = grid_block + 2 * D.G;
        D.ain = grid_block + 3 * D.G; D.amix = grid_block + 4 * D.G; D.aout = grid_block + 5 * D.G;
        HIP_TRY(hipMalloc((void**)&d_nhits, 8 * sizeof(int)));         // [0] hit counter (even frames, backward pass), [1] ncand, [2..3] last counts, [4] hit counter of odd frames
        HIP_TRY(hipMemsetAsync(d_nhits, 0, 8 * sizeof(int), stream));
        HIP_TRY(hipMalloc((void**)&d_hits, (size_t)D.Npad * sizeof(Hit)));
        HIP_TRY(hipMalloc((void**)&d_hits2, (size_t)D.Npad * sizeof(Hit)));
        HIP_TRY(hipMalloc((void**)&vdrift, D.G * sizeof(Vec4<R>)));
        HIP_TRY(hipMemsetAsync(vdrift, 0, D.G * sizeof(Vec4<R>), stream));
        D.vdrift = vdrift;
        // tail reduction (DevSim::tail_on): one arrival counter per grid block, zero between launches.  SMAC_TAIL_REDUCE=0: k_grid_op / the reduction launch as in round 4.
        D.tail_on = 0; D.tail_extra = 0; D.tail_expect = nullptr;
        D.tail_rule = tail_env ? 1 : 0;
        D.hits_next = d_hits2;
        D.zero_next_hits = 0;
        D.keep_vmix = 0;
        D.halo_hs.count = 0; D.halo_hs.slot[0] = D.halo_hs.slot[1] = 0; D.halo_hs.plane0[0] = D.halo_hs.plane0[1] = 0;
        D.halo_np = 0; D.halo_send = nullptr; D.halo_recv = nullptr;
        HIP_TRY(hipMalloc((void**)&d_pmask, (size_t)D.Npad * sizeof(int)));
        D.nhits = d_nhits;
        D.nhits_next = d_nhits + 4;
        D.last_counts = d_nhits + 2;
        D.ncand = d_nhits + 1;
        D.hits = d_hits;
        D.pmask = d_pmask;
        const int Pn = c.n_primitives > 0 ? c.n_primitives : 1;
        const size_t ps = (size_t)Pn * c.max_frames * 13 * sizeof(double);          // primitive state is f64 in both modes
        HIP_TRY(hipMalloc((void**)&D.prim_state, ps));
        HIP_TRY(hipMalloc((void**)&D.prim_grad, ps));
        HIP_TRY(hipMemsetAsync(D.prim_state, 0, ps, stream));
        HIP_TRY(hipMemsetAsync(D.prim_grad, 0, ps, stream));
        HIP_TRY(hipMalloc((void**)&D.ext_f, 2 * Pn * 6 * sizeof(double)));
        HIP_TRY(hipMemsetAsync(D.ext_f, 0, 2 * Pn * 6 * sizeof(double), stream));
        D.ext_f_grad = D.ext_f + Pn * 6;
        const size_t ab = (size_t)Pn * c.max_frames * 6 * sizeof(double);
        HIP_TRY(hipMalloc((void**)&action_buf, ab));
        HIP_TRY(hipMalloc((void**)&action_buf_grad, ab));
        HIP_TRY(hipMemsetAsync(action_buf, 0, ab, stream));
        HIP_TRY(hipMemsetAsync(action_buf_grad, 0, ab, stream));
        const int nc = c.n_control > 0 ? c.n_control : 1;
        HIP_TRY(hipMalloc((void**)&D.action, nc * 3 * sizeof(R)));
        HIP_TRY(hipMalloc((void**)&D.action_grad, nc * 3 * sizeof(R)));
        HIP_TRY(hipMemsetAsync(D.action, 0, nc * 3 * sizeof(R), stream));
        HIP_TRY(hipMemsetAsync(D.action_grad, 0, nc * 3 * sizeof(R), stream));
        HIP_TRY(hipMalloc((void**)&d_control_idx, D.Npad * sizeof(int)));
        HIP_TRY(hipMemsetAsync(d_control_idx, 0, D.Npad * sizeof(int), stream));   // ti.field default 0 (:76)
        D.control_idx = d_control_idx;
        HIP_TRY(hipMalloc((void**)&d_counter, sizeof(unsigned long long)));
        // block-sparse grid + sort scratch
        D.open_x = (c.flags >> 1) & 3;                               // flags bits 1,2: neighbour slab at the low / high x end
        D.nb = c.n_grid / 4;
        nblocks = D.nb * D.nb * D.nb;
        HIP_TRY(hipMalloc((void**)&d_tail_cnt, ((size_t)nblocks + 1) * sizeof(int)));
        HIP_TRY(hipMemsetAsync(d_tail_cnt, 0, ((size_t)nblocks + 1) * sizeof(int), stream));
        D.tail_cnt = d_tail_cnt;
        sort_interval = c.sort_interval > 0 ? c.sort_interval : 80;      // (32 until the wide tiles of round 4: a particle that crosses a block face no longer costs its wave the slow path;
                                                                         //  40 until the end of round 5: the fused kernels' times are flat over 160 substeps of S-grip, profiles/r05_sort_interval.txt)
        HIP_TRY(hipMalloc((void**)&d_cell_count, D.G * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&d_vmax, sizeof(unsigned)));
        HIP_TRY(hipMalloc((void**)&d_vmax_part, ((size_t)D.Npad / 64 + 8) * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&d_over_prefix, D.G * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&d_bin_mask, (size_t)nblocks * KMAX * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc((void**)&d_bin, ((size_t)nblocks * KMAX + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&d_bin_start, ((size_t)nblocks * KMAX + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&d_key, D.Npad * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&d_slot, D.Npad * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&d_dest, D.Npad * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&d_map, D.Npad * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&d_block_start, (nblocks + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&d_guest, 4 * (size_t)(nblocks + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&d_block_chunks, (nblocks + 1) * sizeof(int)));
        HIP_TRY(hipMemsetAsync(d_block_chunks, 0, (nblocks + 1) * sizeof(int), stream));        // (entry [nblocks] stays 0: the scans read one element past the blocks)
        HIP_TRY(hipMalloc((void**)&d_chunk_start, (nblocks + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&d_active_flag, (nblocks + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&d_active_start, (nblocks + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&tmp_frame, fs));
        // [0]: a particle out-ran its binning; [1]: a contact hit list did not fit its checkpoint slot; [2]: a particle left its slab's shared planes
        HIP_TRY(hipMalloc((void**)&d_drift, 4 * sizeof(int)));
        HIP_TRY(hipMemsetAsync(d_drift, 0, 4 * sizeof(int), stream));
        HIP_TRY(hipHostMalloc((void**)&h_nhits, (size_t)c.max_frames * sizeof(int), hipHostMallocDefault));
        for (int i = 0; i < c.max_frames; ++i) h_nhits[i] = -1;
        HIP_TRY(hipHostMalloc((void**)&h_sort_info, 16 * sizeof(int), hipHostMallocDefault));
        memset(h_sort_info, 0, 16 * sizeof(int));
        D.slab_base_lo = 1; D.slab_base_hi = 0;                        // no slab range check until smac_comm_slab / smac_set_slab_range says so
        D.drift_flag = d_drift;
        epochs.clear();
        epochs.emplace_back();
        epochs[0].live = true;
        frame_epoch.assign(c.max_frames, -1);
        adj_epoch.assign(c.max_frames, -1);
        adj_stale.assign(c.max_frames, 0);
        ck_epoch.assign(c.max_frames, -1);
        ck_has_hits.assign(c.max_frames, 0);
        ck_gen.assign(c.max_frames, -1);
        ck_enabled = c.grad_enabled && !(getenv("SMAC_NO_CHECKPOINT") && atoi(getenv("SMAC_NO_CHECKPOINT")));
        if (c.flags & 1) ck_enabled = false;                    // bit 0 of flags: recompute in substep_grad like the reference
        for (int i = 0; i < SMAC_MAX_PRIMS; ++i) {
            D.prim[i].sdf = nullptr; D.prim[i].normal = nullptr; D.prim[i].contact = 0;
            D.prim[i].friction = (R)0.9; D.prim[i].softness = (R)666.0; D.prim[i].inv_dx = (R)1;
            for (int d = 0; d < 3; ++d) { D.prim[i].res[d] = 2; D.prim[i].lower[d] = 0; D.prim[i].upper[d] = 0; }
            D.prim64[i].sdf = nullptr; D.prim64[i].normal = nullptr; D.prim64[i].contact = 0;
            D.prim64[i].friction = 0.9; D.prim64[i].softness = 666.0; D.prim64[i].inv_dx = 1.0;
            for (int d = 0; d < 3; ++d) { D.prim64[i].res[d] = 2; D.prim64[i].lower[d] = 0; D.prim64[i].upper[d] = 0; }
        }
        HIP_TRY(hipStreamSynchronize(stream));
        return SMAC_OK;
    }

    // ---- profiling helpers
    hipEvent_t get_event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        hipEventCreate(&e);
        return e;
    }
    void prof_begin(int id) {
        if (!profiling) return;
        Rec r{id, get_event(), get_event()};
        hipEventRecord(r.a, stream);
        recs.push_back(r);
    }
    void prof_end() {
        if (!profiling) return;
        hipEventRecord(recs.back().b, stream);
    }
    int prof_collect() {
        if (recs.empty()) return SMAC_OK;
        HIP_TRY(hipStreamSynchronize(stream));
        for (auto& r : recs) {
            float ms = 0;
            hipEventElapsedTime(&ms, r.a, r.b);
            prof_ms[r.id] += ms; prof_n[r.id] += 1;
            pool.push_back(r.a); pool.push_back(r.b);
        }
        recs.clear();
        return SMAC_OK;
    }
    int check_launch() {
        HIP_TRY(hipGetLastError());
        return SMAC_OK;
    }

    int sync() override { HIP_TRY(hipStreamSynchronize(stream)); return check_drift(); }

    // ---- IO -----------------------------------------------------------------------------
    // host copy of an epoch's slot -> original-id table (identity for epoch 0)
    // The conversion (f64 AOS in the caller's order <-> R rows in the frame's order) runs on the device: the host only
    // copies the caller's array (1M particles: reset 89 -> 30 ms).
    double* d_io = nullptr;
    size_t io_cap = 0;
    int io_buffer(size_t doubles) {
        if (doubles > io_cap) {
            hipFree(d_io);
            d_io = nullptr;
            HIP_TRY(hipMalloc((void**)&d_io, doubles * sizeof(double)));
            io_cap = doubles;
        }
        return SMAC_OK;
    }
    int upload_comp(R* base, int f, int c0, int cnt, const double* src, int kind, int e) {     // kind: k_rows_from_aos `ident`
        int rc;
        if ((rc = io_buffer((size_t)D.N * cnt))) return rc;
        HIP_TRY(hipMemcpyAsync(d_io, src, (size_t)D.N * cnt * sizeof(double), hipMemcpyHostToDevice, stream));
        R* d = base + (size_t)f * frame_scalars() + rowbase(c0, D.Npad);
        hipLaunchKernelGGL(k_rows_from_aos<R>, dim3(nblk(D.Npad)), dim3(BLOCK), 0, stream, D.N, D.Npad, (const double*)d_io, cnt, 0, cnt,
                           e > 0 ? (const int*)epochs[e].orig : (const int*)nullptr, kind, d);
        HIP_TRY(hipStreamSynchronize(stream));
        return check_launch();
    }
    int download_comp(const R* base, int f, int c0, int cnt, double* dst, int kind, int e) {
        int rc;
        if ((rc = io_buffer((size_t)D.N * cnt))) return rc;
        const R* s = base + (size_t)f * frame_scalars() + rowbase(c0, D.Npad);
        hipLaunchKernelGGL(k_rows_to_aos<R>, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, D.Npad, s, cnt,
                           e > 0 ? (const int*)epochs[e].orig : (const int*)nullptr, kind, d_io, cnt, 0);
        HIP_TRY(hipMemcpyAsync(dst, d_io, (size_t)D.N * cnt * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return check_launch();
    }
    int check_frame(int f) {
        REQUIRE(f >= 0 && f < cfg.max_frames, "frame index out of range");
        return SMAC_OK;
    }

    int set_frame(int f, const double* x, const double* v, const double* F, const double* C) override {
        int rc = check_frame(f);
        if (rc || (rc = leave_fused_forward())) return rc;
        if (frame_epoch[f] < 0) frame_epoch[f] = 0;          // first write: identity order
        if (x && v && F && C) frame_epoch[f] = 0;            // a full overwrite needs no old order (and the particle SET may be new: migration)
        ck_epoch[f] = -1;                                    // the saved forward grid of this frame is stale
        const int e = frame_epoch[f];
        if (x && (rc = upload_comp(D.S, f, CX, 3, x, 2, e))) return rc;
        if (v && (rc = upload_comp(D.S, f, CV, 3, v, 0, e))) return rc;
        if (C && (rc = upload_comp(D.S, f, CC, 9, C, 0, e))) return rc;
        if (F && (rc = upload_comp(D.S, f, CF, 9, F, 1, e))) return rc;
        return SMAC_OK;
    }
    int get_state24(int f, double* out) override {                            // mpm_simulator.py:541-548 get_state: (N, 24) = x3 v3 F9 C9
        int rc = check_frame(f);
        if (rc || (rc = check_drift())) return rc;
        REQUIRE(out, "null argument");
        const int e = frame_epoch[f] < 0 ? 0 : frame_epoch[f];
        if ((rc = io_buffer((size_t)D.N * 24))) return rc;
        const R* fr = D.S + (size_t)f * frame_scalars();
        const int* orig = e > 0 ? (const int*)epochs[e].orig : (const int*)nullptr;
        auto rows = [&](int c0, int cnt, int offset, int ident) {
            hipLaunchKernelGGL(k_rows_to_aos<R>, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, D.Npad, fr + rowbase(c0, D.Npad), cnt, orig, ident,
                               d_io, 24, offset);
        };
        rows(CX, 3, 0, 2); rows(CV, 3, 3, 0); rows(CF, 9, 6, 1); rows(CC, 9, 15, 0);
        HIP_TRY(hipMemcpyAsync(out, d_io, (size_t)D.N * 24 * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return check_launch();
    }
    int get_frame(int f, double* x, double* v, double* F, double* C) override {
        int rc = check_frame(f);
        if (rc || (rc = check_drift())) return rc;
        const int e = frame_epoch[f] < 0 ? 0 : frame_epoch[f];
        if (x && (rc = download_comp(D.S, f, CX, 3, x, 2, e))) return rc;
        if (v && (rc = download_comp(D.S, f, CV, 3, v, 0, e))) return rc;
        if (C && (rc = download_comp(D.S, f, CC, 9, C, 0, e))) return rc;
        if (F && (rc = download_comp(D.S, f, CF, 9, F, 1, e))) return rc;
        return SMAC_OK;
    }
    int reset(const double* state, int cols) override {                       // mpm_simulator.py:494-519
        REQUIRE(state && (cols == 3 || cols == 24), "reset: cols must be 3 or 24");
        int rc;
        p2g_done_frame = 0;                                                   // (a new episode: D.vdrift all zero whatever an aborted one left on blocks nobody swept)
        if ((rc = leave_fused_forward())) return rc;
        frame_epoch[0] = 0;                                                   // user data: identity order, re-binned at the next substep
        ck_epoch[0] = -1;
        // the caller's (N, cols) array goes up once; the split into x / v / C / F rows (host layout x3 v3 F9 C9) happens on the device
        if ((rc = io_buffer((size_t)D.N * cols))) return rc;
        HIP_TRY(hipMemcpyAsync(d_io, state, (size_t)D.N * cols * sizeof(double), hipMemcpyHostToDevice, stream));
        R* fr = D.S;
        auto rows = [&](int c0, int cnt, int offset, int ident) {
            hipLaunchKernelGGL(k_rows_from_aos<R>, dim3(nblk(D.Npad)), dim3(BLOCK), 0, stream, D.N, D.Npad, (const double*)d_io, cols, offset,
                               cnt, (const int*)nullptr, ident, fr + rowbase(c0, D.Npad));
        };
        if (cols != 24) HIP_TRY(hipMemsetAsync(fr, 0, frame_scalars() * sizeof(R), stream));   // v = 0, C = 0, F = I (stored as E = 0)
        rows(CX, 3, 0, 2);
        if (cols == 24) rows(CV, 3, 3, 0);
        if ((rc = check_launch())) return rc;
        // bin the particles now (positions and speeds are on the device) and write every row from the caller's array in the binned order
        if (!getenv("SMAC_RESET_UNSORTED")) return sort_frame(0, false, false, d_io, cols);
        if (cols == 24) {
            rows(CF, 9, 6, 1);
            rows(CC, 9, 15, 0);
        }
        HIP_TRY(hipStreamSynchronize(stream));
        return check_launch();
    }
    int copy_frame(int src, int dst) override {                               // :468-479
        int rc;
        if ((rc = check_frame(src)) || (rc = check_frame(dst)) || (rc = leave_fused_forward())) return rc;
        if (src == dst) return SMAC_OK;
        HIP_TRY(hipMemcpyAsync(D.S + (size_t)dst * frame_scalars(), D.S + (size_t)src * frame_scalars(),
                               frame_scalars() * sizeof(R), hipMemcpyDeviceToDevice, stream));
        frame_epoch[dst] = frame_epoch[src];
        ck_epoch[dst] = -1;
        ++config_gen;                                         // primitive frames were copied too
        for (int i = 0; i < D.P; ++i)
            for (int j = 0; j < cfg.substeps; ++j) {
                if (src + j >= cfg.max_frames || dst + j >= cfg.max_frames) break;
                double* base = D.prim_state + (size_t)i * cfg.max_frames * 13;
                HIP_TRY(hipMemcpyAsync(base + (size_t)(dst + j) * 13, base + (size_t)(src + j) * 13, 13 * sizeof(double),
                                       hipMemcpyDeviceToDevice, stream));
                double* ab = action_buf + (size_t)i * cfg.max_frames * 6;
                HIP_TRY(hipMemcpyAsync(ab + (size_t)(dst + j) * 6, ab + (size_t)(src + j) * 6, 6 * sizeof(double),
                                       hipMemcpyDeviceToDevice, stream));
            }
        if (D.cloth.present) {                               // soft_cloth copyframe :597-602: contact face, penetration flag, the sheet's frames
            const ClothDev& C = D.cloth;
            hipLaunchKernelGGL(k_cloth_copy_ids, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, (const int*)(C.contact_id + (size_t)src * C.n_ids),
                               C.contact_id + (size_t)dst * C.n_ids, (const signed char*)(C.penetration + (size_t)src * C.n_ids),
                               C.penetration + (size_t)dst * C.n_ids);
            const size_t b = (size_t)C.V * 3;
            for (int j = 0; j < cfg.substeps; ++j) {
                if (src + j >= cfg.max_frames || dst + j >= cfg.max_frames) break;
                HIP_TRY(hipMemcpyAsync(C.pos + (dst + j) * b, C.pos + (src + j) * b, b * sizeof(double), hipMemcpyDeviceToDevice, stream));
                HIP_TRY(hipMemcpyAsync(C.vel + (dst + j) * b, C.vel + (src + j) * b, b * sizeof(double), hipMemcpyDeviceToDevice, stream));
            }
        }
        return SMAC_OK;
    }
    int need_grad() {
        REQUIRE(D.A != nullptr, "handle created with grad_enabled = 0");
        return SMAC_OK;
    }
    // ---- adjoint frame storage: one slot per frame (the reference's layout), or a rolling pool (smac_config.adjoint_frames)
    int adj_slots = 0;
    std::vector<int> adj_slot, free_slots;
    std::vector<char> adj_evicted;
    bool rolling() const { return adj_slots < cfg.max_frames; }
    R* adj_ptr(int f) {                                    // nullptr: pool exhausted (rolling mode only)
        if (!rolling()) return D.A + (size_t)f * frame_scalars();
        if (adj_slot[f] < 0) {
            if (free_slots.empty()) return nullptr;
            adj_slot[f] = free_slots.back();
            free_slots.pop_back();
            adj_epoch[f] = -1;                              // a fresh slot is a logically-zero frame whose memory is stale
            adj_stale[f] = 1;
            adj_evicted[f] = 0;
        }
        return D.A + (size_t)adj_slot[f] * frame_scalars();
    }
    void adj_release(int f) {
        if (!rolling() || f < 0 || f >= cfg.max_frames || adj_slot[f] < 0) return;
        free_slots.push_back(adj_slot[f]);
        adj_slot[f] = -1;
        adj_epoch[f] = -1;
        adj_stale[f] = 0;
        adj_evicted[f] = 1;
    }
    static constexpr const char* kPoolMessage =
        "rolling adjoint storage exhausted: smac_config.adjoint_frames must cover the frames a loss has seeded and the backward "
        "sweep has not reached yet, plus 3";
    // clear_grads() is lazy: an adjoint frame is only physically zeroed if somebody is about to read it
    // or add to it.  A backward sweep overwrites A[f] (write mode of g2p_grad / p2g_grad), so it never is.
    int adj_make_zero(int f) {
        R* a = adj_ptr(f);
        REQUIRE(a, kPoolMessage);
        if (!adj_stale[f]) return SMAC_OK;
        HIP_TRY(hipMemsetAsync(a, 0, frame_scalars() * sizeof(R), stream));
        adj_stale[f] = 0;
        return SMAC_OK;
    }
    int get_grad(int f, double* gx, double* gv, double* gF, double* gC) override {
        int rc;
        if ((rc = need_grad()) || (rc = check_frame(f)) || (rc = check_drift())) return rc;
        if (rolling() && adj_slot[f] < 0) {
            REQUIRE(!adj_evicted[f], "get_grad: this adjoint frame has been released (rolling adjoint storage, smac_config.adjoint_frames)");
            if (gx) memset(gx, 0, (size_t)D.N * 3 * sizeof(double));       // never seeded, never reached: zero
            if (gv) memset(gv, 0, (size_t)D.N * 3 * sizeof(double));
            if (gC) memset(gC, 0, (size_t)D.N * 9 * sizeof(double));
            if (gF) memset(gF, 0, (size_t)D.N * 9 * sizeof(double));
            return SMAC_OK;
        }
        if (adj_epoch[f] < 0 && (rc = adj_make_zero(f))) return rc;
        const int e = adj_epoch[f] < 0 ? 0 : adj_epoch[f];
        const R* base = adj_ptr(f);                              // download_comp adds f * frame size to its base: pass f = 0
        if (gx && (rc = download_comp(base, 0, CX, 3, gx, 0, e))) return rc;
        if (gv && (rc = download_comp(base, 0, CV, 3, gv, 0, e))) return rc;
        if (gC && (rc = download_comp(base, 0, CC, 9, gC, 0, e))) return rc;
        if (gF && (rc = download_comp(base, 0, CF, 9, gF, 0, e))) return rc;
        return SMAC_OK;
    }
    int add_grad(int f, const double* gx, const double* gv, const double* gF, const double* gC) override {
        int rc;
        if ((rc = need_grad()) || (rc = check_frame(f))) return rc;
        if (adj_epoch[f] < 0 && (rc = adj_make_zero(f))) return rc;
        if (adj_epoch[f] < 0) adj_epoch[f] = frame_epoch[f] < 0 ? 0 : frame_epoch[f];   // empty adjoint frame: adopt the state's order
        const int e = adj_epoch[f];
        const double* src[4] = {gx, gv, gC, gF};
        const int c0[4] = {CX, CV, CC, CF}, cnt[4] = {3, 3, 9, 9};
        for (int a = 0; a < 4; ++a) {                                         // `x.grad[f, i] += ...` on the device
            if (!src[a]) continue;
            if ((rc = io_buffer((size_t)D.N * cnt[a]))) return rc;
            HIP_TRY(hipMemcpyAsync(d_io, src[a], (size_t)D.N * cnt[a] * sizeof(double), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(k_rows_add_aos<R>, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, D.Npad, (const double*)d_io, cnt[a],
                               e > 0 ? (const int*)epochs[e].orig : (const int*)nullptr,
                               adj_ptr(f) + rowbase(c0[a], D.Npad));
            HIP_TRY(hipStreamSynchronize(stream));                            // d_io is reused by the next component
        }
        return SMAC_OK;
    }
    // `x.grad[f, i] += ...` from arrays that already lie in device memory (a loss evaluated on the GPU, as the reference's loss kernels are:
    // losses/loss_pour.py:130-140): no staging copy, no synchronisation
    int add_grad_device(int f, const double* gx, const double* gv, const double* gF, const double* gC) override {
        int rc;
        if ((rc = need_grad()) || (rc = check_frame(f))) return rc;
        if (adj_epoch[f] < 0 && (rc = adj_make_zero(f))) return rc;
        if (adj_epoch[f] < 0) adj_epoch[f] = frame_epoch[f] < 0 ? 0 : frame_epoch[f];   // empty adjoint frame: adopt the state's order
        const int e = adj_epoch[f];
        const double* src[4] = {gx, gv, gC, gF};
        const int c0[4] = {CX, CV, CC, CF}, cnt[4] = {3, 3, 9, 9};
        for (int a = 0; a < 4; ++a) {
            if (!src[a]) continue;
            hipPointerAttribute_t at;
            REQUIRE(hipPointerGetAttributes(&at, src[a]) == hipSuccess && at.type == hipMemoryTypeDevice, "add_grad_device: not a device pointer");
            hipLaunchKernelGGL(k_rows_add_aos<R>, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, D.Npad, src[a], cnt[a],
                               e > 0 ? (const int*)epochs[e].orig : (const int*)nullptr, adj_ptr(f) + rowbase(c0[a], D.Npad));
        }
        return check_launch();
    }
    int clear_grads() override {
        if (D.A) adj_stale.assign(cfg.max_frames, 1);                              // zeroed on demand (adj_make_zero)
        adj_epoch.assign(cfg.max_frames, -1);
        if (D.A && rolling()) {
            for (int f = 0; f < cfg.max_frames; ++f) adj_release(f);
            adj_evicted.assign(cfg.max_frames, 0);
            adj_stale.assign(cfg.max_frames, 0);
        }
        const int Pn = D.P > 0 ? D.P : 1;
        HIP_TRY(hipMemsetAsync(D.prim_grad, 0, (size_t)Pn * cfg.max_frames * 13 * sizeof(double), stream));
        HIP_TRY(hipMemsetAsync(action_buf_grad, 0, (size_t)Pn * cfg.max_frames * 6 * sizeof(double), stream));
        HIP_TRY(hipMemsetAsync(D.ext_f_grad, 0, Pn * 6 * sizeof(double), stream));
        HIP_TRY(hipMemsetAsync(D.action_grad, 0, (D.n_control > 0 ? D.n_control : 1) * 3 * sizeof(R), stream));
        HIP_TRY(hipMemsetAsync(grid_block + 3 * D.G, 0, 3 * D.G * sizeof(Vec4<R>), stream));
        if (D.cloth.present && D.cloth.pos_grad) {
            const size_t vs = (size_t)cfg.max_frames * D.cloth.V * 3 * sizeof(double);
            HIP_TRY(hipMemsetAsync(D.cloth.pos_grad, 0, vs, stream));
            HIP_TRY(hipMemsetAsync(D.cloth.vel_grad, 0, vs, stream));
            HIP_TRY(hipMemsetAsync(D.cloth.ext_f_grad, 0, (size_t)D.cloth.V * 3 * sizeof(double), stream));
        }
        return SMAC_OK;
    }
    // clear_grads, except that the particle adjoint of frame `src` survives as the adjoint of frame `dst`, order tag included (windowed episodes)
    int carry_grad(int src, int dst) override {
        int rc;
        if ((rc = need_grad()) || (rc = check_frame(src)) || (rc = check_frame(dst))) return rc;
        REQUIRE(!rolling(), "carry_grad: not with rolling adjoint storage (smac_config.adjoint_frames)");
        REQUIRE(g2p_done_frame < 0, "carry_grad: a batched backward sweep is in flight");
        const int e = adj_epoch[src];
        if (e < 0) return clear_grads();                                      // frame `src` carries no adjoint: all zero
        if (!tmp_frame2) HIP_TRY(hipMalloc((void**)&tmp_frame2, frame_scalars() * sizeof(R)));
        HIP_TRY(hipMemcpyAsync(tmp_frame2, adj_ptr(src), frame_scalars() * sizeof(R), hipMemcpyDeviceToDevice, stream));
        if ((rc = clear_grads())) return rc;
        HIP_TRY(hipMemcpyAsync(adj_ptr(dst), tmp_frame2, frame_scalars() * sizeof(R), hipMemcpyDeviceToDevice, stream));
        adj_epoch[dst] = e;                                                   // (keeps that epoch alive: gc_epochs looks at the adjoint frames too)
        adj_stale[dst] = 0;
        return SMAC_OK;
    }
    int set_control_idx(const int32_t* idx) override {
        REQUIRE(idx, "null idx");
        std::vector<int> tmp(D.Npad, -1);
        for (int p = 0; p < D.N; ++p) tmp[p] = D.n_control == 0 ? 0 : idx[p];          // :599-602
        HIP_TRY(hipMemcpyAsync(d_control_idx, tmp.data(), D.Npad * sizeof(int), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SMAC_OK;
    }
    unsigned char* d_mat_id = nullptr;
    int mat2_set = 0;                   // bits: mu2, lam2, yield_ratio2 have been set (smac_set_param)
    int set_material_ids(const int32_t* ids) override {                       // per-particle selector of the two-entry material table (nullptr: one material again)
        if (!ids) { D.mat_id = nullptr; ++config_gen; return SMAC_OK; }
        REQUIRE(D.collision_type != CONTACT_PARTICLE || !any_contact(), "set_material_ids: not with penalty contact (collision_type 1)");
        // entry 1 must have been described: an entry left at its zero initialisation has no stiffness and - plastic, von Mises - yields at zero stress
        // (ADVICE r4).  The reference fills yield_stress uniformly from the config (mpm_simulator.py:86-90): a missing yield_ratio2 means the SAME yield stress.
        REQUIRE((mat2_set & 3) == 3, "set_material_ids: set entry 1 first (smac_set_param \"mu2\" and \"lam2\")");
        if (!(mat2_set & 4)) D.yield_c2 = D.mu2 > R(0) ? D.mat.yield_c * D.mat.mu / D.mu2 : D.mat.yield_c;
        std::vector<unsigned char> tmp(D.Npad, 0);
        for (int p = 0; p < cfg.n_particles; ++p) {
            REQUIRE(ids[p] == 0 || ids[p] == 1, "set_material_ids: entries must be 0 or 1");
            tmp[p] = (unsigned char)ids[p];
        }
        if (!d_mat_id) HIP_TRY(hipMalloc((void**)&d_mat_id, D.Npad));
        HIP_TRY(hipMemcpyAsync(d_mat_id, tmp.data(), D.Npad, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        D.mat_id = d_mat_id;
        ++config_gen;                                    // the forward grids on file were made with the old materials
        return SMAC_OK;
    }
    int dense_grid_m(int f) {                                                 // dense row-major grid_m of frame f in dense_tmp
        if (!dense_tmp) HIP_TRY(hipMalloc((void**)&dense_tmp, D.G * sizeof(R)));
        HIP_TRY(hipMemsetAsync(dense_tmp, 0, D.G * sizeof(R), stream));
        const R* Sf = D.S + (size_t)f * frame_scalars();
        hipLaunchKernelGGL(k_grid_m_only<R>, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, Sf, Sf + rowbase(1, D.Npad), Sf + rowbase(2, D.Npad), D.N,
                           D.n, D.inv_dx, D.p_mass, dense_tmp);
        return check_launch();
    }
    int compute_grid_m(int f, double* out) override {
        int rc = check_frame(f);
        if (rc) return rc;
        if ((rc = dense_grid_m(f))) return rc;
        if (out) {
            std::vector<R> tmp(D.G);
            HIP_TRY(hipMemcpyAsync(tmp.data(), dense_tmp, D.G * sizeof(R), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            for (size_t i = 0; i < D.G; ++i) out[i] = (double)tmp[i];
        }
        return SMAC_OK;
    }
    // ---- chamfer loss (smac_loss.hpp) -------------------------------------------------------
    struct PIdx { int *cell_start = nullptr, *count = nullptr, *key = nullptr, *ids = nullptr, *slots = nullptr; double* pts = nullptr; int cap = 0, npts = 0; };
    PIdx pi_target, pi_cur;
    double *d_target = nullptr, *d_loss = nullptr;
    int n_target = 0;
    int pi_res() const { return (D.n + 7) / 8 * 8; }
    int pi_alloc(PIdx& I, int m) {
        const size_t cells = (size_t)pi_res() * pi_res() * pi_res();
        if (!I.cell_start) {
            HIP_TRY(hipMalloc((void**)&I.cell_start, (cells + 1) * sizeof(int)));
            HIP_TRY(hipMalloc((void**)&I.count, (cells + 1) * sizeof(int)));
        }
        if (m > I.cap) {
            hipFree(I.key); hipFree(I.ids); hipFree(I.slots); hipFree(I.pts);
            HIP_TRY(hipMalloc((void**)&I.key, (size_t)m * sizeof(int)));
            HIP_TRY(hipMalloc((void**)&I.ids, (size_t)m * sizeof(int)));
            HIP_TRY(hipMalloc((void**)&I.slots, (size_t)m * sizeof(int)));
            HIP_TRY(hipMalloc((void**)&I.pts, (size_t)m * 3 * sizeof(double)));
            I.cap = m;
        }
        return SMAC_OK;
    }
    // counting sort of m points (rows x0..x2 of a frame, or an (m,3) f64 array) into the two-level cell order
    int pi_build(PIdx& I, int m, const R* x0, const R* x1, const R* x2, const double* aos, const int* orig) {
        int rc;
        if ((rc = pi_alloc(I, m))) return rc;
        const int n = pi_res();
        const size_t cells = (size_t)n * n * n;
        HIP_TRY(hipMemsetAsync(I.count, 0, (cells + 1) * sizeof(int), stream));
        hipLaunchKernelGGL(k_pi_count<R>, dim3(nblk(m)), dim3(BLOCK), 0, stream, m, x0, x1, x2, aos, n, I.count, I.key);
        if ((rc = scan(I.count, I.cell_start, (int)cells + 1))) return rc;
        HIP_TRY(hipMemsetAsync(I.count, 0, (cells + 1) * sizeof(int), stream));
        hipLaunchKernelGGL(k_pi_fill<R>, dim3(nblk(m)), dim3(BLOCK), 0, stream, m, x0, x1, x2, aos, (const int*)I.key,
                           (const int*)I.cell_start, I.count, orig, I.ids, I.slots, I.pts);
        I.npts = m;
        return check_launch();
    }
    PointIndex pi_view(const PIdx& I) const { PointIndex v = {pi_res(), I.npts, I.cell_start, I.ids, I.slots, I.pts}; return v; }
    int loss_set_target(const double* target, int m) override {                  // loss_pour.py:29-31 load_target_position
        REQUIRE(target && m > 0, "loss_set_target: empty target");
        hipFree(d_target);
        d_target = nullptr;
        HIP_TRY(hipMalloc((void**)&d_target, (size_t)m * 3 * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(d_target, target, (size_t)m * 3 * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        n_target = m;
        return pi_build(pi_target, m, nullptr, nullptr, nullptr, d_target, nullptr);
    }
    int loss_chamfer(int f, double weight, int add_grad, double* loss_out) override {   // loss_pour.py:44-70 (+ its adjoint)
        int rc;
        if ((rc = check_frame(f)) || (rc = check_drift())) return rc;
        REQUIRE(n_target > 0, "loss_chamfer: no target set (smac_loss_set_target)");
        REQUIRE(frame_epoch[f] >= 0, "loss_chamfer: frame holds no state");
        const int e = frame_epoch[f];
        const R* Sf = D.S + (size_t)f * frame_scalars();
        if ((rc = pi_build(pi_cur, D.N, Sf, Sf + rowbase(1, D.Npad), Sf + rowbase(2, D.Npad), nullptr, e > 0 ? (const int*)epochs[e].orig : nullptr))) return rc;
        R* Af = nullptr;
        if (add_grad && (rc = adjoint_frame_for_seeding(f, &Af))) return rc;
        if (!d_loss) HIP_TRY(hipMalloc((void**)&d_loss, sizeof(double)));
        HIP_TRY(hipMemsetAsync(d_loss, 0, sizeof(double), stream));
        R* g0 = Af; R* g1 = Af ? Af + rowbase(1, D.Npad) : nullptr; R* g2 = Af ? Af + rowbase(2, D.Npad) : nullptr;
        hipLaunchKernelGGL(k_chamfer_cur_to_target<R>, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, Sf, Sf + rowbase(1, D.Npad),
                           Sf + rowbase(2, D.Npad), pi_view(pi_target), weight, add_grad ? 1 : 0, g0, g1, g2, d_loss);
        hipLaunchKernelGGL(k_chamfer_target_to_cur<R>, dim3(nblk(n_target)), dim3(BLOCK), 0, stream, n_target, (const double*)d_target,
                           pi_view(pi_cur), weight, add_grad ? 1 : 0, g0, g1, g2, d_loss);
        double h = 0;
        HIP_TRY(hipMemcpyAsync(&h, d_loss, sizeof h, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (loss_out) *loss_out = h;
        return check_launch();
    }
    // adjoint frame f made addressable in the particle order of state frame f (zeroed / re-ordered as needed)
    int adjoint_frame_for_seeding(int f, R** Af) {
        int rc;
        if ((rc = need_grad())) return rc;
        const int e = frame_epoch[f];
        if (adj_epoch[f] < 0) {
            if ((rc = adj_make_zero(f))) return rc;
        } else if (adj_epoch[f] != e) {
            const R* tmp = nullptr;
            if ((rc = adjoint_in_order(f, e, &tmp))) return rc;
            HIP_TRY(hipMemcpyAsync(adj_ptr(f), tmp, frame_scalars() * sizeof(R), hipMemcpyDeviceToDevice, stream));
        }
        adj_epoch[f] = e;
        *Af = adj_ptr(f);
        REQUIRE(*Af, kPoolMessage);
        return SMAC_OK;
    }
    unsigned long long* d_best = nullptr;
    double* d_md_out = nullptr;
    int loss_min_dist(int f, int id0, int id1, const double* c3, double offset, double weight, int add_grad, double* out4) override {
        int rc;
        if ((rc = check_frame(f)) || (rc = check_drift())) return rc;
        REQUIRE(c3 && out4, "null argument");
        REQUIRE(frame_epoch[f] >= 0, "loss_min_dist: frame holds no state");
        REQUIRE(id0 >= 0 && id0 < id1 && id1 <= D.N, "loss_min_dist: bad particle range");
        const int e = frame_epoch[f];
        const R* Sf = D.S + (size_t)f * frame_scalars();
        R* Af = nullptr;
        if (add_grad && (rc = adjoint_frame_for_seeding(f, &Af))) return rc;
        if (!d_best) {
            HIP_TRY(hipMalloc((void**)&d_best, sizeof(unsigned long long)));
            HIP_TRY(hipMalloc((void**)&d_md_out, 4 * sizeof(double)));
        }
        HIP_TRY(hipMemsetAsync(d_best, 0xff, sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(k_min_dist<R>, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, Sf, Sf + rowbase(1, D.Npad), Sf + rowbase(2, D.Npad),
                           e > 0 ? (const int*)epochs[e].orig : (const int*)nullptr, id0, id1, c3[0], c3[1], c3[2], offset, d_best);
        hipLaunchKernelGGL(k_min_dist_finish<R>, dim3(1), dim3(64), 0, stream, Sf, Sf + rowbase(1, D.Npad), Sf + rowbase(2, D.Npad),
                           (const unsigned long long*)d_best, c3[0], c3[1], c3[2], offset, weight, add_grad ? 1 : 0, Af,
                           Af ? Af + rowbase(1, D.Npad) : (R*)nullptr, Af ? Af + rowbase(2, D.Npad) : (R*)nullptr, d_md_out);
        HIP_TRY(hipMemcpyAsync(out4, d_md_out, 4 * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return check_launch();
    }
    // particles inside a contact band / chunks holding one, as left by the most recent forward substep
    int contact_counts(int32_t* nhits, int32_t* nchunks_hit) override {
        int h[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(h, d_nhits + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (nhits) *nhits = h[0];
        if (nchunks_hit) *nchunks_hit = h[1];
        return SMAC_OK;
    }
    int count_active_cells(int f, int64_t* cells) override {
        int rc = check_frame(f);
        if (rc) return rc;
        if ((rc = dense_grid_m(f))) return rc;
        HIP_TRY(hipMemsetAsync(d_counter, 0, sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(k_count_active<R>, dim3(nblk(D.G)), dim3(BLOCK), 0, stream, (const R*)dense_tmp, D.G, d_counter);
        if ((rc = check_launch())) return rc;
        unsigned long long h = 0;
        HIP_TRY(hipMemcpyAsync(&h, d_counter, sizeof h, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        *cells = (int64_t)h;
        return SMAC_OK;
    }

    // ---- epochs / sorting ------------------------------------------------------------------
    static void free_epoch(Epoch& e) {
        hipFree(e.orig); hipFree(e.cellrank); hipFree(e.from_prev); hipFree(e.inv); hipFree(e.chunks); hipFree(e.active); hipFree(e.block_chunk_start); hipFree(e.block_chunks); hipFree(e.block_active);
        hipFree(e.block_slot); hipFree(e.tail_expect);
        e.tail_expect = nullptr;
        e.orig = e.cellrank = e.from_prev = e.inv = e.active = e.block_chunk_start = e.block_chunks = e.block_active = e.block_slot = nullptr;
        e.chunks = nullptr;
        e.live = false;
    }
    // a dropped epoch hands its device buffers to the pool (kernels still using them are ahead of any new writer on the stream)
    void retire_epoch(Epoch& e) {
        Epoch b;
        b.orig = e.orig; b.cellrank = e.cellrank; b.from_prev = e.from_prev; b.inv = e.inv; b.chunks = e.chunks; b.active = e.active;
        b.block_chunk_start = e.block_chunk_start; b.block_chunks = e.block_chunks; b.block_active = e.block_active;
        b.block_slot = e.block_slot;
        b.tail_expect = e.tail_expect;
        epoch_pool.push_back(std::move(b));
        e.orig = e.cellrank = e.from_prev = e.inv = e.active = e.block_chunk_start = e.block_chunks = e.block_active = e.block_slot = e.tail_expect = nullptr;
        e.chunks = nullptr;
        e.inv_valid = false;
        e.live = false;
    }
    size_t chunk_capacity() const { const int n = cfg.n_particles; return (size_t)n / 256 + (size_t)(nblocks < n ? nblocks : n) + 8; }
    int epoch_buffers(Epoch& ep) {
        if (!epoch_pool.empty()) {
            Epoch b = std::move(epoch_pool.back());
            epoch_pool.pop_back();
            ep.orig = b.orig; ep.cellrank = b.cellrank; ep.from_prev = b.from_prev; ep.inv = b.inv; ep.chunks = b.chunks; ep.active = b.active;
            ep.block_chunk_start = b.block_chunk_start; ep.block_chunks = b.block_chunks; ep.block_active = b.block_active;
            ep.block_slot = b.block_slot;
            ep.tail_expect = b.tail_expect;
            ep.inv_valid = false;
            return SMAC_OK;
        }
        HIP_TRY(hipMalloc((void**)&ep.orig, D.Npad * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&ep.cellrank, D.Npad * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&ep.from_prev, D.Npad * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&ep.chunks, chunk_capacity() * sizeof(Chunk)));
        HIP_TRY(hipMalloc((void**)&ep.active, ((size_t)nblocks + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&ep.block_chunk_start, (nblocks + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&ep.block_chunks, (nblocks + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&ep.block_active, (nblocks + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&ep.block_slot, (nblocks + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&ep.tail_expect, (nblocks + 1) * sizeof(int)));
        return SMAC_OK;
    }
    void gc_epochs() {                                   // drop epochs no frame refers to any more
        std::vector<char> used(epochs.size(), 0);
        used[0] = 1;
        used[grid_epoch] = 1;
        for (int e : frame_epoch) if (e > 0) used[e] = 1;
        for (int e : adj_epoch) if (e > 0) used[e] = 1;
        for (size_t e = 1; e < epochs.size(); ++e)
            if (!used[e] && epochs[e].live) retire_epoch(epochs[e]);
    }
    int new_epoch_slot() {
        for (size_t e = 1; e < epochs.size(); ++e)
            if (!epochs[e].live) return (int)e;
        epochs.emplace_back();
        return (int)epochs.size() - 1;
    }
    int scan(const int* in, int* out, int n) {
        size_t need = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, need, in, out, n, stream);
        if (need > cub_bytes) {
            hipFree(d_cub);
            HIP_TRY(hipMalloc(&d_cub, need));
            cub_bytes = need;
        }
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_cub, need, in, out, n, stream));
        return SMAC_OK;
    }
    // Re-bin frame f (smac_sort.hpp).  The frame is rewritten in the new order and gets a new epoch.
    // ---- a particle out-ran its binning (k_g2p / the scatter kernels raised the drift flag): instead of failing, the frames of the
    // epoch in use are recomputed from its first frame - which is intact and freshly binned - with a re-sort before every substep.
    // What cannot be replayed stays an error: the slab phases (neighbours would have to replay too), particle actions (one buffer,
    // already overwritten by later env steps) and a backward pass that has already consumed the bad frames.
    int fwd_head = -1;                 // frame the forward pass has reached (f + 1 of the last substep)
    bool repairing = false, bwd_since_fwd = false, slab_phase_used = false;
    double* ext_snap = nullptr;        // ext_f at frame `snap_frame`: the start of the current epoch, or the last point at which the host
    double* cloth_ext_snap = nullptr;  // cleared a wrench accumulator inside it (env-step boundary) - the replay re-accumulates from THERE
    int snap_frame = -1;
    int replay_count_from = -1;        // repair_drift: substeps before this frame replay with their wrench sums diverted to the scratch slot
    int snapshot_ext(int at_frame) {
        snap_frame = at_frame;
        const int Pn = D.P > 0 ? D.P : 1;
        if (!ext_snap) HIP_TRY(hipMalloc((void**)&ext_snap, Pn * 6 * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(ext_snap, D.ext_f, Pn * 6 * sizeof(double), hipMemcpyDeviceToDevice, stream));
        if (D.cloth.present) {
            if (!cloth_ext_snap) HIP_TRY(hipMalloc((void**)&cloth_ext_snap, (size_t)D.cloth.V * 3 * sizeof(double)));
            HIP_TRY(hipMemcpyAsync(cloth_ext_snap, D.cloth.ext_f, (size_t)D.cloth.V * 3 * sizeof(double), hipMemcpyDeviceToDevice, stream));
        }
        return SMAC_OK;
    }
    int repair_drift(int f_end, int e_bad) {
        // (the cloth variant: its contact-face search and penetration tracing run on the host's schedule between the substeps - the calls made on every
        //  frame of the epoch are on file, cloth_ops_of_frame, and are made again on the recomputed frame; round 4)
        if (repairing || bwd_since_fwd || slab_phase_used || e_bad <= 0 || !epochs[e_bad].live || !ext_snap || (D.cloth.present && !cloth_ext_snap))
            return SMAC_ERR_INVALID;
        const int fs = epochs[e_bad].frame;
        if (!(fs < f_end) || frame_epoch[fs] != e_bad) return SMAC_ERR_INVALID;
        repairing = true;
        const int Pn = D.P > 0 ? D.P : 1;
        int rc = SMAC_OK;
        p2g_done_frame = 0;                                   // (forces the reset below: a particle that out-ran its binning may have added to D.vdrift on blocks
        if (leave_fused_forward() != SMAC_OK) rc = SMAC_ERR_HIP;   //  no k_grid_op of this epoch sweeps)
        // The wrench accumulators go back to their value at `snap_frame` - the epoch's first frame, or the last env-step boundary inside it at which
        // the host read and cleared ext_f (rigid_simulator.py:92-93, 117): what the host has consumed must not be accumulated again.  Substeps
        // replayed BEFORE that frame send their wrench sums to the scratch slot (as the backward recompute does); from that frame on they count.
        replay_count_from = snap_frame < fs ? fs : snap_frame;
        if (hipMemcpyAsync(D.ext_f, ext_snap, Pn * 6 * sizeof(double), hipMemcpyDeviceToDevice, stream) != hipSuccess) rc = SMAC_ERR_HIP;
        if (D.cloth.present && hipMemcpyAsync(D.cloth.ext_f, cloth_ext_snap, (size_t)D.cloth.V * 3 * sizeof(double), hipMemcpyDeviceToDevice, stream) != hipSuccess)
            rc = SMAC_ERR_HIP;
        const int keep = sort_interval;
        sort_interval = 1;
        epochs[e_bad].interval = 1;                          // frame fs keeps its (fresh) binning; every later frame is re-binned before it is used
        const std::vector<double> action_keep = action_now;
        for (int g = fs; g < f_end && !rc; ++g) {
            // the action frame g's substep ran with (particle controllers); action.grad is a backward quantity and no backward substep has run since
            const bool has = D.n_control > 0 && g < (int)action_of_frame.size() && !action_of_frame[g].empty();
            rc = substep_phase(g, has ? action_of_frame[g].data() : nullptr, -1);
            // the host's calls on the frame this substep produced, in their order (soft_cloth/engine/taichi_env.py:95-98: get_contact_pair, trace_penetration...)
            if (!rc && D.cloth.present && g + 1 < (int)cloth_ops_of_frame.size())
                for (size_t i = 0; i < cloth_ops_of_frame[g + 1].size() && !rc; ++i) rc = cloth_contact_run(cloth_ops_of_frame[g + 1][i], g + 1);
        }
        if (!rc && D.n_control > 0 && !action_keep.empty()) rc = set_action(action_keep.data());
        replay_count_from = -1;
        sort_interval = keep;
        repairing = false;
        ++drift_repairs;
        return rc;
    }
    int drift_repairs = 0;

    // `aos` (reset): the frame's x and v rows are already there in the caller's order; after the binning ALL rows are written straight from the
    // caller's (N, cols) array in the new order (192 contiguous bytes per particle) instead of being moved row by row - an episode's first
    // binning then costs no scattered frame move (411 us at 1M particles from a random order) and no copy-back.
    int resorts_done = 0;
    long long epoch_serial = 0;
    int stable_ranks = getenv("SMAC_STABLE_RANKS") ? atoi(getenv("SMAC_STABLE_RANKS")) : 1;   // 0: every re-sort hands the ranks out afresh (round 2)
    int guests_env = getenv("SMAC_GUESTS") ? atoi(getenv("SMAC_GUESTS")) : 1;                  // 0: every particle is binned in its own block (rounds 1 - 5)
    int sort_frame(int f, bool read_drift = false, bool allow_repair = true, const double* aos = nullptr, int cols = 0) {
        const int e_old = frame_epoch[f];
        gc_epochs();
        const int e_new = new_epoch_slot();
        prof_begin(K_SORT);
        const int nbins = nblocks * KMAX;
        R* Sf = D.S + (size_t)f * frame_scalars();
        HIP_TRY(hipMemsetAsync(d_cell_count, 0, D.G * sizeof(int), stream));
        HIP_TRY(hipMemsetAsync(d_bin + nbins, 0, sizeof(int), stream));
        // guests (smac_sort.hpp): blocks hand their few particles beyond whole chunks to a face neighbour with room - fewer chunks, the unit the fused kernels pay for.
        // Not with a sheet (its contact-face search culls the faces per chunk by the chunk's block); SMAC_GUESTS=0: off.  (The slab pieces take it too: the planes a
        // rank exchanges are grid nodes, whichever chunk a particle that touches them sits in.)
        const bool guests = guests_env && !D.cloth.present && D.N > 0;
        int *g_count = d_guest, *g_in = d_guest + (nblocks + 1), *g_dir = d_guest + 2 * (size_t)(nblocks + 1), *g_left = d_guest + 3 * (size_t)(nblocks + 1);
        if (guests) {
            HIP_TRY(hipMemsetAsync(d_guest, 0, 2 * (size_t)(nblocks + 1) * sizeof(int), stream));
            hipLaunchKernelGGL(k_block_count<R>, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, (const R*)Sf, (const R*)(Sf + rowbase(1, D.Npad)),
                               (const R*)(Sf + rowbase(2, D.Npad)), D.N, D.n, D.nb, g_count);
            hipLaunchKernelGGL(k_donate_plan, dim3(nblk(nblocks)), dim3(BLOCK), 0, stream, nblocks, D.nb, (const int*)g_count, g_in, g_dir, g_left);
        }
        hipLaunchKernelGGL(k_sort_rank<R>, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, (const R*)Sf, (const R*)(Sf + rowbase(1, D.Npad)),
                           (const R*)(Sf + rowbase(2, D.Npad)), (const R*)(Sf + rowbase(3, D.Npad)), (const R*)(Sf + rowbase(4, D.Npad)),
                           (const R*)(Sf + rowbase(5, D.Npad)), D.N, D.n, D.nb, D.inv_dx, d_cell_count, d_key, d_slot, d_vmax_part,
                           (const int*)((e_old > 0 && stable_ranks && D.G <= ((size_t)1 << 26)) ? epochs[e_old].cellrank : nullptr),
                           guests ? (const int*)g_dir : (const int*)nullptr, guests ? g_left : (int*)nullptr);
        hipLaunchKernelGGL(k_bin_masks, dim3((nblocks + 3) / 4), dim3(BLOCK), 0, stream, nblocks, (const int*)d_cell_count, d_bin,
                           d_bin_mask, d_over_prefix, (const float*)d_vmax_part, nblk(D.N), (float*)d_vmax);
        int rc = scan(d_bin, d_bin_start, nbins + 1);
        if (rc) return rc;
        Epoch ep;
        if ((rc = epoch_buffers(ep))) return rc;
        hipLaunchKernelGGL(k_sort_dest, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, (const int*)d_key, (const int*)d_slot, (const int*)d_cell_count,
                           (const int*)d_bin_start, (const unsigned long long*)d_bin_mask, (const int*)d_over_prefix,
                           (const int*)(e_old > 0 ? epochs[e_old].orig : nullptr), ep.from_prev, ep.orig, D.G <= ((size_t)1 << 26) ? ep.cellrank : (int*)nullptr);
        int* const d_dest = ep.from_prev;                          // (the destination map is written where the epoch keeps it: no copy)
        last_dest = d_dest;
        ep.prev = e_old;
        ep.prev_serial = e_old > 0 ? epochs[e_old].serial : -1;
        ep.serial = ++epoch_serial;
        if (aos) {
            auto rows = [&](int c0, int cnt, int offset, int ident) {
                hipLaunchKernelGGL(k_rows_from_aos<R>, dim3(nblk(D.Npad)), dim3(BLOCK), 0, stream, D.N, D.Npad, aos, cols, offset, cnt, (const int*)ep.orig,
                                   ident, Sf + rowbase(c0, D.Npad));
            };
            if (cols != 24) HIP_TRY(hipMemsetAsync(Sf, 0, frame_scalars() * sizeof(R), stream));   // v = 0, C = 0, F = I (stored as E = 0)
            rows(CX, 3, 0, 2);
            if (cols == 24) { rows(CV, 3, 3, 0); rows(CF, 9, 6, 1); rows(CC, 9, 15, 0); }
        } else {
            hipLaunchKernelGGL(k_sort_move<R>, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, (const int*)d_dest, (const R*)Sf, tmp_frame,
                               D.Npad, (int)NCOMP);
            HIP_TRY(hipMemcpyAsync(Sf, tmp_frame, frame_scalars() * sizeof(R), hipMemcpyDeviceToDevice, stream));
        }
        ++resorts_done;
        // block info, chunk list, active list
        HIP_TRY(hipMemsetAsync(d_active_flag, 0, (nblocks + 1) * sizeof(int), stream));      // (d_block_chunks: k_block_info writes every entry, [nblocks] stays 0)
        hipLaunchKernelGGL(k_block_info, dim3(nblk(nblocks)), dim3(BLOCK), 0, stream, nblocks, D.nb, (const int*)d_bin_start, D.N,
                           d_block_start, d_block_chunks, d_active_flag);
        if ((rc = scan(d_block_chunks, d_chunk_start, nblocks + 1))) return rc;
        if ((rc = scan(d_active_flag, d_active_start, nblocks + 1))) return rc;
        // the chunk and active lists and the epoch's per-block tables are written BEFORE the host has the totals (k_emit_lists guards the chunk list's
        // capacity itself), and the host gets what it needs - totals, the fastest speed, the ending epoch's drift flags - in one write to pinned memory:
        // after the synchronisation nothing of the re-sort is left to enqueue
        hipLaunchKernelGGL(k_emit_lists, dim3(nblk(nblocks + 1)), dim3(BLOCK), 0, stream, nblocks, D.N, (const int*)d_bin_start,
                           (const int*)d_block_start, (const int*)d_block_chunks, (const int*)d_chunk_start,
                           (const int*)d_active_flag, (const int*)d_active_start, ep.chunks, ep.active, (int)chunk_capacity(),
                           ep.block_chunk_start, ep.block_chunks, ep.block_active, ep.block_slot);
        hipLaunchKernelGGL(k_tail_expect, dim3(nblk(nblocks)), dim3(BLOCK), 0, stream, nblocks, D.nb, (const int*)d_block_chunks, ep.tail_expect);
        hipLaunchKernelGGL(k_sort_info, dim3(1), dim3(64), 0, stream, (const int*)(d_chunk_start + nblocks), (const int*)(d_active_start + nblocks),
                           (const unsigned*)d_vmax, (const int*)d_drift, read_drift ? 1 : 0, h_sort_info);
        HIP_TRY(hipStreamSynchronize(stream));
        const int totals[2] = {((volatile int*)h_sort_info)[0], ((volatile int*)h_sort_info)[1]};
        const unsigned vbits = (unsigned)((volatile int*)h_sort_info)[2];
        int flags2[4] = {((volatile int*)h_sort_info)[3], ((volatile int*)h_sort_info)[4], ((volatile int*)h_sort_info)[5], ((volatile int*)h_sort_info)[6]};
        // a drift error of the epoch that ends here is reported AFTER the new epoch is committed: S[f] is already in the
        // new order, so frame_epoch[f] must name it or every later get_state / set_frame of this frame would be scrambled
        const int drifted = flags2[0];
        if (flags2[0] || flags2[1] || flags2[2]) HIP_TRY(hipMemsetAsync(d_drift, 0, 4 * sizeof(int), stream));
        {
            // A particle may move 4 cells (its block's halo) before the binning breaks; budget 2 cells for the fastest
            // particle at its current speed, which leaves a factor two for acceleration inside the interval.
            float vmax;
            memcpy(&vmax, &vbits, sizeof vmax);
            const double travel = (double)vmax * cfg.dt * D.n;                   // cells per substep
            int iv = sort_interval;
            if (travel > 0 && 2.0 / travel < (double)iv) iv = (int)(2.0 / travel);
            ep.interval = iv < 1 ? 1 : iv;
        }
        ep.nchunks = totals[0];
        ep.nactive = totals[1];
        if ((size_t)ep.nchunks > chunk_capacity()) {
            epoch_pool.push_back(std::move(ep));                                  // (device buffers go back to the pool)
            err = "internal: chunk list capacity exceeded";
            return SMAC_ERR_INVALID;
        }
        prof_end();
        ep.frame = f;
        ep.live = true;
        epochs[e_new] = std::move(ep);
        frame_epoch[f] = e_new;
        if ((size_t)epochs[e_new].nchunks > slab_chunks) {
            hipFree(slab);
            hipFree(d_cand);
            slab_chunks = (size_t)epochs[e_new].nchunks + epochs[e_new].nchunks / 8 + 16;
            HIP_TRY(hipMalloc((void**)&slab, slab_chunks * TILE_WORDS * sizeof(Vec4<R>)));
            HIP_TRY(hipMalloc((void**)&d_cand, slab_chunks * sizeof(int)));
        }
        if (flags2[1] && (rc = hit_overflow())) return rc;
        REQUIRE(!flags2[2], kSlabLeftMessage);
        if (drifted) {
            if (allow_repair && !aos && repair_drift(f, e_old) == SMAC_OK) return sort_frame(f, true, false);   // frame f is recomputed: bin it again
            err = kDriftMessage;
            return SMAC_ERR_INVALID;
        }
        // (inside a replay, frames before `replay_count_from` do not own the accumulator's value: the snapshot stays where it is)
        if (!(repairing && f < replay_count_from) && (rc = snapshot_ext(f))) return rc;
        return check_launch();
    }
    // make epoch e the one the kernels see; the grid blocks the previous epoch may have dirtied are zeroed
    int bind_epoch(int e) {
        if (e != grid_epoch && grid_epoch > 0 && epochs[grid_epoch].live && epochs[grid_epoch].nactive > 0) {
            DevSim<R> Do = D;
            Do.active = epochs[grid_epoch].active;
            Do.nactive = epochs[grid_epoch].nactive;
            hipLaunchKernelGGL(k_clear_active<R>, dim3((Do.nactive + 3) / 4), dim3(BLOCK), 0, stream, Do, grid_block, 6);
        }
        // (k_grid_op packs the shared planes of the ACTIVE cells into the send buffer itself: what an earlier epoch's cells left there must read as zero)
        if (e != grid_epoch && sc.on && halo_buf) HIP_TRY(hipMemsetAsync(halo_buf, 0, 2 * halo_records * sizeof(Vec4<R>), stream));
        grid_epoch = e;
        const Epoch& ep = epochs[e];
        D.chunks = ep.chunks; D.nchunks = ep.nchunks; D.active = ep.active; D.nactive = ep.nactive;
        D.orig_id = ep.orig; D.block_chunk_start = ep.block_chunk_start; D.block_chunks = ep.block_chunks;
        D.block_active = ep.block_active;
        D.block_slot = ep.block_slot;
        D.tail_expect = ep.tail_expect;
        D.slab = slab;
        D.cand = d_cand;
        return check_launch();
    }
    int ensure_inverse(int e) {
        Epoch& ep = epochs[e];
        if (!ep.inv) HIP_TRY(hipMalloc((void**)&ep.inv, D.Npad * sizeof(int)));
        if (!ep.inv_valid) {
            hipLaunchKernelGGL(k_invert, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, (const int*)ep.orig, ep.inv);
            ep.inv_valid = true;
        }
        return check_launch();
    }
    // Where the rows of adjoint frame f lie for the particles of epoch `to`, when a table the epochs keep says so without scratch space: the destination
    // map of the re-sort the sweep is crossing, `orig` (frame in identity order) or an inverse table.  The kernels that read frame f + 1's adjoint take
    // such a map (DevSim::An_map) instead of a re-ordered copy of the frame.  *direct = false: SMAC_AN_MAP=0 (adjoint_in_order gathers the frame as before).
    int an_map_env = getenv("SMAC_AN_MAP") ? atoi(getenv("SMAC_AN_MAP")) : 1;
    int adjoint_direct_map(int f, int to, const int** map, bool* direct) {
        *map = nullptr;
        *direct = true;
        const int from = adj_epoch[f];
        if (from < 0 || from == to) return SMAC_OK;                      // same order (all-zero frames have none)
        if (!an_map_env) { *direct = false; return SMAC_OK; }
        if (from > 0 && to > 0 && epochs[from].prev == to && epochs[from].prev_serial == epochs[to].serial && epochs[from].from_prev) *map = epochs[from].from_prev;
        else if (from == 0) *map = epochs[to].orig;
        else if (to == 0) {
            int rc = ensure_inverse(from);
            if (rc) return rc;
            *map = epochs[from].inv;
        } else {
            // two epochs that do not follow one another (a seed stored under an earlier binning of the frame: the loss ran, then the window was simulated again):
            // compose the map once (k_compose, 10 us) and let the kernels read through it - not a gather of the frame on top (44 us).  The composed map lives
            // in scratch of its own: d_map is adjoint_in_order's.
            int rc = ensure_inverse(from);
            if (rc) return rc;
            if (!d_map_an) HIP_TRY(hipMalloc((void**)&d_map_an, D.Npad * sizeof(int)));
            hipLaunchKernelGGL(k_compose, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, (const int*)epochs[to].orig, (const int*)epochs[from].inv, d_map_an);
            *map = d_map_an;
        }
        return SMAC_OK;
    }
    int* d_map_an = nullptr;
    // adjoint frame f re-ordered from its own epoch into epoch `to` -> tmp_frame (returns pointer to use)
    int adjoint_in_order(int f, int to, const R** out, R* dst = nullptr) {
        if (!dst) dst = tmp_frame;
        const R* Af = adj_ptr(f);
        REQUIRE(Af, kPoolMessage);
        const int from = adj_epoch[f];
        if (from < 0 || from == to) { *out = Af; return SMAC_OK; }   // all-zero frames have no order
        int rc;
        const int* map;
        if (from > 0 && to > 0 && epochs[from].prev == to && epochs[from].prev_serial == epochs[to].serial && epochs[from].from_prev)
            map = epochs[from].from_prev;                              // the sweep crosses the re-sort that made `from` out of `to`: its destination map is the gather
        else if (from == 0) map = epochs[to].orig;                     // identity -> sorted: src index = original id
        else {
            if ((rc = ensure_inverse(from))) return rc;
            if (to == 0) map = epochs[from].inv;
            else {
                hipLaunchKernelGGL(k_compose, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, (const int*)epochs[to].orig,
                                   (const int*)epochs[from].inv, d_map);
                map = d_map;
            }
        }
        prof_begin(K_REORDER);
        hipLaunchKernelGGL(k_gather_rows<R>, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, map, Af, dst, D.Npad, (int)NCOMP);
        prof_end();
        *out = dst;
        return check_launch();
    }
    // A substep had more particles inside contact bands than a checkpoint slot holds (max(8192, N/8)): its list was not filed.  The host does
    // not know which frame it was, so every frame on file loses its list and substep_grad repeats the band test for them (k_contact_mask, as it
    // does when the lists do not fit in memory at all) - slower, never wrong.
    int hit_overflows = 0;
    int hit_overflow() {
        std::fill(ck_has_hits.begin(), ck_has_hits.end(), (char)0);
        ++hit_overflows;
        return SMAC_OK;
    }
    // the flags can only change when kernels ran: a second check at the same point of the tape (clear_ext_f / ext_f of the NEXT primitive at an
    // env-step boundary) costs no device round trip
    long long launches_seen = -1, launch_counter = 0;
    int check_drift() {
        if (launches_seen == launch_counter) return SMAC_OK;
        launches_seen = launch_counter;
        int h[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(h_sort_info + 8, d_drift, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));     // (pinned destination: a DMA, not a staged copy)
        HIP_TRY(hipStreamSynchronize(stream));
        for (int i = 0; i < 4; ++i) h[i] = ((volatile int*)h_sort_info)[8 + i];
        if (h[0] || h[1] || h[2]) HIP_TRY(hipMemsetAsync(d_drift, 0, 4 * sizeof(int), stream));
        int rc;
        if (h[1] && (rc = hit_overflow())) return rc;
        REQUIRE(!h[2], kSlabLeftMessage);
        if (h[0]) {
            if (fwd_head > 0 && frame_epoch[fwd_head] > 0 && repair_drift(fwd_head, frame_epoch[fwd_head]) == SMAC_OK) {
                int h2[2] = {0, 0};                           // the replay (re-binning before every substep) must itself come out clean
                HIP_TRY(hipMemcpyAsync(h2, d_drift, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                if (h2[1]) {
                    HIP_TRY(hipMemsetAsync(d_drift + 1, 0, sizeof(int), stream));
                    if ((rc = hit_overflow())) return rc;
                }
                if (!h2[0]) return SMAC_OK;
                HIP_TRY(hipMemsetAsync(d_drift, 0, sizeof(int), stream));
            }
            err = kDriftMessage;
            return SMAC_ERR_INVALID;
        }
        return SMAC_OK;
    }

    // ---- hot path -----------------------------------------------------------------------
    bool any_contact() const {
        if (D.cloth.present) return true;
        for (int i = 0; i < D.P; ++i)
            if (D.prim[i].contact) return true;
        return false;
    }

    // ---- soft <-> cloth contact (soft_cloth/engine/primitive/primitive_cloth.py, soft_cloth/engine/mpm_simulator.py:447-561) ----
    int* d_cloth_faces = nullptr; int* d_cloth_nbr = nullptr; signed char* d_cloth_nbr_dir = nullptr;
    int* d_cloth_warn = nullptr;
    int cloth_pairs_by_chunk = getenv("SMAC_CLOTH_PAIRS_FLAT") ? 0 : 1;   // (the flat kernel, one thread per particle over all faces, is kept for frames in identity order and A/B)
    double* d_cloth_ext_scratch = nullptr;
    int set_param(const char* name, double value) override {
        REQUIRE(name, "set_param: null name");
        if (!strcmp(name, "plasticity")) {
            REQUIRE(value == 0.0 || value == 1.0, "set_param(plasticity): 0 = clipped singular values (softmac), 1 = von Mises (soft_cloth)");
            D.mat.plast = (int)value;
        } else if (!strcmp(name, "yield_ratio")) {
            REQUIRE(value > 0.0, "set_param(yield_ratio): yield_stress / (2 mu) must be positive");
            D.mat.yield_c = (R)value;
        } else if (!strcmp(name, "mu2") || !strcmp(name, "lam2") || !strcmp(name, "yield_ratio2")) {   // entry 1 of the two-entry material table (smac_set_material_ids)
            REQUIRE(value >= 0.0, "set_param(mu2 | lam2 | yield_ratio2): negative");
            (name[0] == 'm' ? D.mu2 : (name[0] == 'l' ? D.lam2 : D.yield_c2)) = (R)value;
            mat2_set |= name[0] == 'm' ? 1 : (name[0] == 'l' ? 2 : 4);
        } else if (!strcmp(name, "mass_eps")) {
            REQUIRE(value >= 0.0, "set_param(mass_eps): negative");
            D.m_eps = (R)value;
        } else if (!strcmp(name, "cloth_pairs_flat")) {
            cloth_pairs_by_chunk = value != 0.0 ? 0 : 1;   // 1: the flat contact-face search (every particle over every face) also on sorted frames
            return SMAC_OK;
        } else if (!strcmp(name, "cloth_hash")) {
            cloth_hash_on = value != 0.0 ? 1 : 0;          // 0: per-chunk search over the whole mesh (round 2), 1: over the broad phase's per-block face lists
            return SMAC_OK;
        } else REQUIRE(false, "set_param: unknown parameter (plasticity | yield_ratio | mu2 | lam2 | yield_ratio2 | mass_eps | cloth_pairs_flat | cloth_hash)");
        ++config_gen;                                   // the forward grids on file were made with the old value
        return SMAC_OK;
    }
    int get_param(const char* name, double* value) override {
        REQUIRE(name && value, "get_param: null argument");
        if (!strcmp(name, "drift_repairs")) *value = (double)drift_repairs;             // epochs recomputed because a particle out-ran its binning
        else if (!strcmp(name, "cloth_hash_entries")) *value = hash_total_host ? (double)*hash_total_host : 0.0;   // (face, block) pairs of the last broad-phase build
        else if (!strcmp(name, "exchanges")) *value = (double)exchanges_done;          // halo exchanges run by smac_substeps_slab[_grad] so far
        else if (!strcmp(name, "comm_world")) *value = (comm || (comm_stub == 2 && ipc.shm)) ? (double)c_world : 0.0;   // ranks of this handle's LIVE communicator (0: none / aborted)
        else if (!strcmp(name, "comm_transport")) *value = comm ? 1.0 : ((comm_stub == 2 && ipc.shm) ? 2.0 : (comm_stub == 1 && sc.on ? 3.0 : 0.0));   // 1 RCCL, 2 IPC test link, 3 device-copy stub, 0 none
        else if (!strcmp(name, "contact_skips")) *value = (double)contact_skips;        // backward substeps that needed no contact adjoint launch (empty filed hit list)
        else if (!strcmp(name, "chunks")) *value = (double)D.nchunks;                  // work items (<= 256 particles of one block each) of the binning in use
        else if (!strcmp(name, "max_hits")) {                                          // the longest contact hit list the checkpoint saves have filed so far (-1: none yet)
            HIP_TRY(hipStreamSynchronize(stream));
            int m = -1;
            for (int i = 0; h_nhits && i < cfg.max_frames; ++i) m = h_nhits[i] > m ? h_nhits[i] : m;
            *value = (double)m;
        }
        else if (!strcmp(name, "hit_overflows")) *value = (double)hit_overflows;      // times a contact hit list did not fit its checkpoint slot (backward then repeats the band test)
        else if (!strcmp(name, "plasticity")) *value = (double)D.mat.plast;
        else if (!strcmp(name, "yield_ratio")) *value = (double)D.mat.yield_c;
        else if (!strcmp(name, "mass_eps")) *value = (double)D.m_eps;
        else if (!strcmp(name, "resorts")) *value = (double)resorts_done;                // re-binnings so far (the first one, at reset, included)
        else if (!strcmp(name, "resort_moved") || !strcmp(name, "resort_far")) {         // particles the LAST re-binning moved to another slot / by more than a chunk (256 slots); counted on request
            unsigned long long h = 0;
            if (resorts_done > 0 && D.N > 0 && last_dest) {
                HIP_TRY(hipMemsetAsync(d_counter, 0, sizeof(unsigned long long), stream));
                hipLaunchKernelGGL(k_count_moved, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, (const int*)last_dest, d_counter, !strcmp(name, "resort_far") ? 256 : 0);
                HIP_TRY(hipMemcpyAsync(&h, d_counter, sizeof(h), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
            }
            *value = (double)h;
        }
        else REQUIRE(false, "get_param: unknown parameter (drift_repairs | hit_overflows | max_hits | chunks | contact_skips | exchanges | comm_world | comm_transport | resorts | resort_moved | resort_far | cloth_hash_entries | plasticity | yield_ratio | mass_eps)");
        return SMAC_OK;
    }
    int cloth_create(int nv, int nf, const int32_t* faces, int nn, const int32_t* nbr, const int8_t* nbr_dir, double friction, double softness,
                     double force_scale, int sticky, double scale) override {
        REQUIRE(!D.cloth.present, "cloth_create: this handle already has a cloth primitive");
        // SDF primitives AND the sheet in one handle (round 4; BASELINE config C5): forecast contact only, on the unit domain - the SDF primitives live on
        // [0,1)^3, the sheet on [0,scale)^3.  The reference has no simulator with both; the chain is primitives in index order, then the sheet (k_contact_hits).
        REQUIRE(D.P == 0 || (cfg.collision_type == CONTACT_MIXED && scale == 1.0),
                "cloth_create: a handle with SDF primitives takes the cloth primitive only with forecast contact (collision_type 2) and mpm_scale 1");
        REQUIRE(nv > 0 && nf > 0 && faces && nn >= 0 && (nn == 0 || (nbr && nbr_dir)) && scale > 0.0, "cloth_create: bad mesh arguments");
        for (int i = 0; i < 3 * nf; ++i) REQUIRE(faces[i] >= 0 && faces[i] < nv, "cloth_create: face index out of range");
        for (size_t i = 0; i < (size_t)nf * nn; ++i) REQUIRE(nbr[i] >= 0 && nbr[i] < nf, "cloth_create: neighbour face index out of range");
        ClothDev& C = D.cloth;
        const size_t vs = (size_t)cfg.max_frames * nv * 3 * sizeof(double), ns = (size_t)cfg.max_frames * cfg.n_particles;
        HIP_TRY(hipMalloc((void**)&d_cloth_faces, (size_t)nf * 3 * sizeof(int)));
        HIP_TRY(hipMemcpy(d_cloth_faces, faces, (size_t)nf * 3 * sizeof(int), hipMemcpyHostToDevice));
        if (nn > 0) {
            HIP_TRY(hipMalloc((void**)&d_cloth_nbr, (size_t)nf * nn * sizeof(int)));
            HIP_TRY(hipMalloc((void**)&d_cloth_nbr_dir, (size_t)nf * nn));
            HIP_TRY(hipMemcpy(d_cloth_nbr, nbr, (size_t)nf * nn * sizeof(int), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d_cloth_nbr_dir, nbr_dir, (size_t)nf * nn, hipMemcpyHostToDevice));
        }
        HIP_TRY(hipMalloc((void**)&C.pos, vs)); HIP_TRY(hipMalloc((void**)&C.vel, vs));
        HIP_TRY(hipMemset(C.pos, 0, vs)); HIP_TRY(hipMemset(C.vel, 0, vs));
        if (cfg.grad_enabled) {
            HIP_TRY(hipMalloc((void**)&C.pos_grad, vs)); HIP_TRY(hipMalloc((void**)&C.vel_grad, vs));
            HIP_TRY(hipMemset(C.pos_grad, 0, vs)); HIP_TRY(hipMemset(C.vel_grad, 0, vs));
        }
        HIP_TRY(hipMalloc((void**)&C.ext_f, (size_t)nv * 3 * sizeof(double))); HIP_TRY(hipMalloc((void**)&C.ext_f_grad, (size_t)nv * 3 * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&d_cloth_ext_scratch, (size_t)nv * 3 * sizeof(double)));
        HIP_TRY(hipMemset(C.ext_f, 0, (size_t)nv * 3 * sizeof(double))); HIP_TRY(hipMemset(C.ext_f_grad, 0, (size_t)nv * 3 * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&C.contact_id, ns * sizeof(int))); HIP_TRY(hipMalloc((void**)&C.penetration, ns));
        HIP_TRY(hipMalloc((void**)&C.contact_before, (size_t)cfg.n_particles * sizeof(int)));
        HIP_TRY(hipMemset(C.contact_id, 0xff, ns * sizeof(int))); HIP_TRY(hipMemset(C.penetration, 0, ns));
        HIP_TRY(hipMemset(C.contact_before, 0xff, (size_t)cfg.n_particles * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&d_cloth_warn, 2 * sizeof(int))); HIP_TRY(hipMemset(d_cloth_warn, 0, 2 * sizeof(int)));
        C.V = nv; C.Fc = nf; C.n_neighbors = nn; C.faces = d_cloth_faces; C.nbr = d_cloth_nbr; C.nbr_dir = d_cloth_nbr_dir;
        C.n_ids = cfg.n_particles;
        C.par.friction = friction; C.par.softness = softness; C.par.force_scale = force_scale; C.par.scale = scale; C.par.sticky = sticky ? 1 : 0;
        C.present = 1;
        ++config_gen;
        return SMAC_OK;
    }
    // ---- broad phase of the contact-face search (smac_cloth_kernels.hpp FaceHash): per grid block the faces whose padded box can hold one of its
    // particles; rebuilt when the sheet's frame or state changed.  The total is read back one call late (no host sync on the substep's path):
    // a list that does not fit raises the device flag, that search scans every face, and the next build has a larger list.
    FaceHash fhash = {nullptr, nullptr, nullptr, 0, nullptr};
    int cloth_hash_on = getenv("SMAC_CLOTH_HASH") ? atoi(getenv("SMAC_CLOTH_HASH")) : 1;
    int hash_frame = -1;
    long long hash_gen = -1, cloth_state_gen = 0;
    int* hash_total_host = nullptr;              // pinned: total list length of the last build
    int cloth_hash_build(int f) {
        const ClothDev& C = D.cloth;
        if (!fhash.count) {
            HIP_TRY(hipMalloc((void**)&fhash.count, ((size_t)nblocks + 1) * sizeof(int)));
            HIP_TRY(hipMalloc((void**)&fhash.start, ((size_t)nblocks + 1) * sizeof(int)));
            HIP_TRY(hipMalloc((void**)&fhash.overflow, sizeof(int)));
            HIP_TRY(hipHostMalloc((void**)&hash_total_host, sizeof(int)));
            *hash_total_host = 0;
        }
        if (hash_frame == f && hash_gen == cloth_state_gen) return SMAC_OK;
        const int want = *hash_total_host > 96 * C.Fc + 4096 ? *hash_total_host + *hash_total_host / 4 : 96 * C.Fc + 4096;   // (the previous build's total: read one call late)
        if (want > fhash.cap) {
            HIP_TRY(hipStreamSynchronize(stream));
            hipFree(fhash.list);
            fhash.list = nullptr;
            HIP_TRY(hipMalloc((void**)&fhash.list, (size_t)want * sizeof(int)));
            fhash.cap = want;
        }
        const int fb = (C.Fc + 255) / 256;
        HIP_TRY(hipMemsetAsync(fhash.count, 0, ((size_t)nblocks + 1) * sizeof(int), stream));
        HIP_TRY(hipMemsetAsync(fhash.overflow, 0, sizeof(int), stream));
        hipLaunchKernelGGL(k_cloth_hash<false>, dim3(fb), dim3(256), 0, stream, C, f, D.n, D.nb, fhash);
        int rc = scan(fhash.count, fhash.start, nblocks + 1);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(hash_total_host, fhash.start + nblocks, sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemsetAsync(fhash.count, 0, ((size_t)nblocks + 1) * sizeof(int), stream));
        hipLaunchKernelGGL(k_cloth_hash<true>, dim3(fb), dim3(256), 0, stream, C, f, D.n, D.nb, fhash);
        hash_frame = f;
        hash_gen = cloth_state_gen;
        return check_launch();
    }
    int cloth_need() { REQUIRE(D.cloth.present, "this handle has no cloth primitive (smac_cloth_create)"); return SMAC_OK; }
    int cloth_set_state(int f0, int f1, const double* pos, const double* vel) override {            // set_all_states :348-354, frames [f0, f1)
        int rc;
        if ((rc = cloth_need())) return rc;
        REQUIRE(pos && vel && f0 >= 0 && f0 < f1 && f1 <= cfg.max_frames, "cloth_set_state: bad frame range / null argument");
        const size_t b = (size_t)D.cloth.V * 3 * sizeof(double);
        for (int f = f0; f < f1; ++f) {
            HIP_TRY(hipMemcpyAsync(D.cloth.pos + (size_t)f * D.cloth.V * 3, pos, b, hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemcpyAsync(D.cloth.vel + (size_t)f * D.cloth.V * 3, vel, b, hipMemcpyHostToDevice, stream));
        }
        HIP_TRY(hipStreamSynchronize(stream));
        for (int f = f0; f < f1; ++f) ck_epoch[f] = -1;   // the sheet's frame f enters substep f only
        ++cloth_state_gen;
        return SMAC_OK;
    }
    int cloth_get_state(int f, double* pos, double* vel, int grad) override {                         // get_all_states :320-324 / get_all_states_grad :334-338
        int rc;
        if ((rc = cloth_need()) || (rc = check_frame(f))) return rc;
        REQUIRE(pos && vel, "null argument");
        REQUIRE(!grad || D.cloth.pos_grad, "the handle was created without gradients");
        const size_t b = (size_t)D.cloth.V * 3 * sizeof(double), at = (size_t)f * D.cloth.V * 3;
        HIP_TRY(hipMemcpyAsync(pos, (grad ? D.cloth.pos_grad : D.cloth.pos) + at, b, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(vel, (grad ? D.cloth.vel_grad : D.cloth.vel) + at, b, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SMAC_OK;
    }
    int cloth_ext_f(int op, double* buf) override {        // 0: ext_f.to_numpy(); 1: clear_ext_f :285-290; 2: set_ext_f_grad :292-296
        int rc;
        if ((rc = cloth_need())) return rc;
        const size_t b = (size_t)D.cloth.V * 3 * sizeof(double);
        if (op == 0) {
            REQUIRE(buf, "null argument");
            if ((rc = check_drift())) return rc;                                      // a drifted epoch is repaired BEFORE the host consumes the sheet's force
            HIP_TRY(hipMemcpyAsync(buf, D.cloth.ext_f, b, hipMemcpyDeviceToHost, stream));
        } else if (op == 1) {
            if (fwd_head >= 0 && !bwd_since_fwd && (rc = check_drift())) return rc;       // (repairs with the accumulator as the forward pass left it)
            HIP_TRY(hipMemsetAsync(D.cloth.ext_f, 0, b, stream));
            HIP_TRY(hipMemsetAsync(D.cloth.ext_f_grad, 0, b, stream));
            if (ext_snap && fwd_head >= 0 && !repairing && (rc = snapshot_ext(fwd_head))) return rc;   // a later repair re-accumulates from here
        } else if (op == 2) {
            REQUIRE(buf, "null argument");
            HIP_TRY(hipMemcpyAsync(D.cloth.ext_f_grad, buf, b, hipMemcpyHostToDevice, stream));
        } else REQUIRE(false, "cloth_ext_f: unknown op");
        HIP_TRY(hipStreamSynchronize(stream));
        return SMAC_OK;
    }
    // id -> slot table of frame f (nullptr: the frame is in identity order)
    int frame_inverse(int f, const int** out) {
        const int e = frame_epoch[f];
        *out = nullptr;
        if (e <= 0) return SMAC_OK;
        int rc;
        if ((rc = ensure_inverse(e))) return rc;
        *out = epochs[e].inv;
        return SMAC_OK;
    }
    // 0: get_contact_pair :447-469; 1: backup_contact_pair :471-482; 2: trace_penetration_after_mpm :484-518; 3: ..._after_cloth :520-553
    // the calls the host made on each frame since its last forward substep produced it (op codes, in order): what a drift repair makes again
    std::vector<std::vector<signed char>> cloth_ops_of_frame;
    int cloth_contact(int op, int f) override {
        int rc;
        if ((rc = cloth_need()) || (rc = check_frame(f)) || (rc = check_drift())) return rc;
        REQUIRE(op >= 0 && op <= 3, "cloth_contact: unknown op");
        if ((int)cloth_ops_of_frame.size() < cfg.max_frames) cloth_ops_of_frame.resize(cfg.max_frames);
        if (cloth_ops_of_frame[f].size() >= 16) cloth_ops_of_frame[f].erase(cloth_ops_of_frame[f].begin());   // (frame 0 is never produced by a substep: bounded)
        cloth_ops_of_frame[f].push_back((signed char)op);
        return cloth_contact_run(op, f);
    }
    int cloth_contact_run(int op, int f) {
        int rc;
        REQUIRE(frame_epoch[f] >= 0, "cloth contact: frame holds no particle state");
        const ClothDev& C = D.cloth;
        const size_t at = (size_t)f * C.n_ids;
        const R* Sf = D.S + (size_t)f * frame_scalars();
        if (op == 0) {
            const int e = frame_epoch[f];
            if (e > 0 && cloth_pairs_by_chunk) {              // sorted frame: per-chunk face culling, faces from the broad phase's per-block lists
                if ((rc = bind_epoch(e))) return rc;
                if (cloth_hash_on && (rc = cloth_hash_build(f))) return rc;
                if (cloth_hash_on) hipLaunchKernelGGL((k_cloth_pairs_chunk<R, true>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, D, f, Sf, fhash);
                else hipLaunchKernelGGL((k_cloth_pairs_chunk<R, false>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, D, f, Sf, fhash);
            } else
                hipLaunchKernelGGL(k_cloth_pairs<R>, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D, f, Sf, e > 0 ? (const int*)epochs[e].orig : (const int*)nullptr);
        } else if (op == 1) {
            hipLaunchKernelGGL(k_cloth_copy_ids, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, (const int*)(C.contact_id + at), C.contact_before,
                               (const signed char*)nullptr, (signed char*)nullptr);
        } else if (op == 2 || op == 3) {
            REQUIRE(f >= 1 && frame_epoch[f - 1] >= 0, "trace_penetration: needs frame f-1");
            const int *ic = nullptr, *ip = nullptr;
            if ((rc = frame_inverse(f, &ic)) || (rc = frame_inverse(f - 1, &ip))) return rc;
            hipLaunchKernelGGL(k_cloth_trace<R>, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D, f, op == 3 ? 1 : 0, Sf, Sf - frame_scalars(), ic, ip, d_cloth_warn);
        } else REQUIRE(false, "cloth_contact: unknown op");
        ck_epoch[f] = -1;                               // the contact faces / penetration flags of frame f are inputs of THAT frame's forward grid only
        return check_launch();                          // (a global invalidation made every slab-phase backward pass fail: its checkpoints are mandatory)
    }
    int cloth_get_contact(int f, int32_t* ids, int8_t* pen) override {
        int rc;
        if ((rc = cloth_need()) || (rc = check_frame(f))) return rc;
        const size_t at = (size_t)f * D.cloth.n_ids;
        if (ids) HIP_TRY(hipMemcpyAsync(ids, D.cloth.contact_id + at, (size_t)D.N * sizeof(int), hipMemcpyDeviceToHost, stream));
        if (pen) HIP_TRY(hipMemcpyAsync(pen, D.cloth.penetration + at, (size_t)D.N, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SMAC_OK;
    }
    int cloth_set_contact(int f, const int32_t* ids, const int8_t* pen) override {            // reset_all_kernel :642-643; reset_kernel :629
        int rc;
        if ((rc = cloth_need()) || (rc = check_frame(f))) return rc;
        const size_t at = (size_t)f * D.cloth.n_ids;
        if (ids) {
            for (int i = 0; i < D.N; ++i) REQUIRE(ids[i] >= -1 && ids[i] < D.cloth.Fc, "cloth_set_contact: face id out of range");
            HIP_TRY(hipMemcpyAsync(D.cloth.contact_id + at, ids, (size_t)D.N * sizeof(int), hipMemcpyHostToDevice, stream));
        }
        if (pen) HIP_TRY(hipMemcpyAsync(D.cloth.penetration + at, pen, (size_t)D.N, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        ck_epoch[f] = -1;
        if (f < (int)cloth_ops_of_frame.size()) cloth_ops_of_frame[f].clear();      // (the host's values stand: a drift repair does not search this frame again)
        return SMAC_OK;
    }
    int cloth_check_penetration(int f, int32_t* total, int32_t* warnings) override {           // check_penetration :555-561
        int rc;
        if ((rc = cloth_need()) || (rc = check_frame(f))) return rc;
        REQUIRE(total, "null argument");
        HIP_TRY(hipMemsetAsync(d_cloth_warn + 1, 0, sizeof(int), stream));
        hipLaunchKernelGGL(k_cloth_count_pen, dim3(nblk(D.N)), dim3(BLOCK), 0, stream, D.N, (const signed char*)(D.cloth.penetration + (size_t)f * D.cloth.n_ids),
                           d_cloth_warn + 1);
        int h[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(h, d_cloth_warn, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        *total = h[1];
        if (warnings) *warnings = h[0];               // particles whose previous contact face was not among the neighbours (tracing skipped them)
        return check_launch();
    }
    int set_segment(int n_live, int frame_shift) override {
        REQUIRE(n_live >= 1 && n_live <= cfg.n_particles, "set_segment: n_live must be in [1, n_particles (the handle's capacity)]");
        REQUIRE(frame_shift >= 0, "set_segment: negative frame_shift");
        D.N = n_live;
        D.frame_shift = frame_shift;
        return SMAC_OK;
    }
    int set_action_v(const double* action) override {
        REQUIRE(action, "null action");
        return set_action(action);
    }
    // The particle action in force (host copy) and, per frame, the one its forward substep ran with: a drift repair replays an epoch that may span
    // several env steps, each with its own action (round 4: before, scenes with particle controllers reported a drifted epoch as an error)
    std::vector<double> action_now;
    std::vector<std::vector<double>> action_of_frame;
    int set_action(const double* action) {                                    // :579-592
        REQUIRE(D.n_control > 0, "action given but n_control == 0");
        REQUIRE(D.n_control <= 64, "n_control > 64");
        action_now.assign(action, action + 3 * D.n_control);
        // the values travel in the kernel's argument block (copied at launch): no staging buffer whose lifetime a stream sync would have to guard
        SmallArgs<R, 3 * 64> a;
        for (int i = 0; i < 3 * D.n_control; ++i) a.v[i] = (R)action[i];
        hipLaunchKernelGGL((k_set_small<R, 3 * 64>), dim3(1), dim3(256), 0, stream, a, 3 * D.n_control, D.action, D.action_grad);   // action.grad zeroed (:584-586)
        return check_launch();
    }
    int stage_ext_f_grad(const double* ext_f_grad) {                          // :342-344 set_ext_f_grad of every primitive, without a host sync
        if (!ext_f_grad || D.P <= 0) return SMAC_OK;
        SmallArgs<double, SMAC_MAX_PRIMS * 6> a;
        for (int i = 0; i < 6 * D.P; ++i) a.v[i] = ext_f_grad[i];
        hipLaunchKernelGGL((k_set_small<double, SMAC_MAX_PRIMS * 6>), dim3(1), dim3(256), 0, stream, a, 6 * D.P, D.ext_f_grad, (double*)nullptr);
        return check_launch();
    }
    int get_action_grad(double* out) override {                               // action.grad.to_numpy() :378
        REQUIRE(out && D.n_control > 0, "get_action_grad: null argument or n_control == 0");
        R tmp[3 * 64];
        HIP_TRY(hipMemcpyAsync(tmp, D.action_grad, 3 * D.n_control * sizeof(R), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (int i = 0; i < 3 * D.n_control; ++i) out[i] = (double)tmp[i];
        return SMAC_OK;
    }
    int check_contact_supported() {
        if (D.P > 0 && any_contact()) {
            for (int i = 0; i < D.P; ++i)
                if (D.prim[i].contact) REQUIRE(D.prim[i].sdf != nullptr, "contact primitive without an uploaded SDF");
        }
        return SMAC_OK;
    }
    // the recompute pass of substep_grad must not double-count ext_f: send its wrench sums to a scratch slot
    double* scratch = nullptr;
    double* scratch_ext() {
        if (!scratch) (void)hipMalloc((void**)&scratch, SMAC_MAX_PRIMS * 6 * sizeof(double));
        return scratch;
    }
    int ngrid_blocks() const { return (D.nactive + 3) / 4; }
    int nchunk_blocks() const { return ((D.nchunks + 7) / 8) * 8; }   // XCD-aware chunk mapping (xcd_chunk) needs a multiple of 8
    // hit list: 8 hits per workgroup and pass; the count lives on the device, so size the grid for the chip (a group
    // that finds no hit left exits at once) - a short grid turns the list into a serial chain of SDF-lookup latencies
    // Workgroups of the two hit-list kernels.  Both walk the list with a grid stride, so ANY count is correct; what the count costs is time: 2,048 workgroups for the
    // 250 that find a hit (2,000 of 1M particles in contact) meant 1,800 empty ones queueing behind the working ones - k_contact_grad holds a whole CU per workgroup -
    // and the launch lasted twice as long as its slowest wave (profiles/r05_contact_grid.txt).  The host cannot read the device's counter without a stall, but the
    // checkpoint saves drop every frame's count into pinned host memory (h_nhits): the count of frame f itself when it has landed (backward: always), else the
    // largest count among the neighbouring frames (the contact set moves slowly), with a quarter on top.  Nothing known (no saves yet): the old 2,048.
    int contact_grid_env = getenv("SMAC_CONTACT_GRID_HINT") ? atoi(getenv("SMAC_CONTACT_GRID_HINT")) : 1;
    int contact_grid_max = getenv("SMAC_CONTACT_GRID_MAX") ? atoi(getenv("SMAC_CONTACT_GRID_MAX")) : 0;      // > 0 (tests): never more workgroups than this - every one walks several rounds
    int contact_grad_grid(int f = -1, bool exact = false) const {
        const int per = (BLOCK / 64) * SMAC_HITS_PER_WAVE, all = (D.N + per - 1) / per;
        int cap = all < 2048 ? (all > 0 ? all : 1) : 2048;
        if (contact_grid_max > 0) return cap < contact_grid_max ? cap : contact_grid_max;
        if (!contact_grid_env || f < 0 || !h_nhits) return cap;
        const volatile int* seen = h_nhits;
        int hint = exact ? seen[f] : -1;
        if (hint < 0) {
            for (int i = f > 24 ? f - 24 : 0; i < cfg.max_frames && i <= f + 24; ++i) hint = seen[i] > hint ? seen[i] : hint;
            if (hint < 0) return cap;
            hint += hint / 4 + 64;
        }
        const int need = (hint + per - 1) / per;
        return need < 8 ? 8 : (need < cap ? need : cap);
    }
    // Forward grid passes.  stage 0: everything; stage 1: clear_grid :93-114 on the active blocks, p2g, (forward
    // kinematics), slab reduction; stage 2: grid_op + contact.  Stages 1/2 exist for the slab decomposition, which
    // sums the {m,p} halo planes across neighbouring GPUs between them.
    // skip_p2g: this substep's P2G already ran inside the G2P launch of the substep before (k_g2p_p2g); fuse_next: the NEXT substep's P2G will ride in this
    // substep's G2P launch - forward_kinematics to frame f + 1 and the emptying of the next hit counter move into k_grid_op's launch
    int forward_grid(int f, bool store_F, bool is_recompute, int stage = 0, bool skip_p2g = false, bool fuse_next = false) {
        int rc;
        if (D.nchunks == 0 || D.nactive == 0) return SMAC_OK;
        D.any_contact = any_contact() ? 1 : 0;
        D.cur_frame = f;
        DevSim<R> Dc = D;                                       // the recompute pass must not double-count ext_f
        if (is_recompute || (repairing && f < replay_count_from)) Dc.ext_f = scratch_ext();
        if ((is_recompute || (repairing && f < replay_count_from)) && D.cloth.present) Dc.cloth.ext_f = d_cloth_ext_scratch;
        const bool fk_in_grid_op = fuse_next && !is_recompute && cfg.rigid_velocity_control && D.P > 0;
        // Tail reduction: the P2G launch itself completes {m,p} and applies grid_op (whole substeps, no grid-node contact: k_grid_op's phase 0 is what the
        // last arriver does).  A P2G that rode in the previous substep's G2P launch says so through p2g_tail_frame.
        const bool can_tail = tail_env && stage == 0 && !(D.collision_type == CONTACT_GRID && D.any_contact);
        bool tail_done = skip_p2g && p2g_tail_frame == f;
        REQUIRE(!tail_done || can_tail, "internal: a tail-reduced P2G in front of a substep that needs k_grid_op's other forms");
        p2g_tail_frame = -1;
        Dc.tail_on = 0; Dc.tail_extra = 0;
        if (stage != 2 && !skip_p2g) {
            if (can_tail) { Dc.tail_on = 1; tail_done = true; }      // (no other reader of {m,p} / v_out runs in a plain P2G launch: tail_extra 0)
            // k_grid_op rewrites {m,p}, v_mixed and v_out of every active cell ({m,p} = slabs + D.vdrift, which it leaves zero again): no clear pass in
            // front of P2G.  The recompute inside substep_grad clears the adjoint fields with it.
            if (is_recompute) {
                prof_begin(K_CLEAR);
                hipLaunchKernelGGL(k_clear_active<R>, dim3(ngrid_blocks()), dim3(BLOCK), 0, stream, D, grid_block, 6);
                prof_end();
            }
            prof_begin(K_P2G);
            const bool pcon = D.collision_type == CONTACT_PARTICLE && D.any_contact;
            if (D.mat_id) {                                     // two-entry material table: its own instantiation (set_material_ids refuses penalty contact)
                REQUIRE(!pcon, "two materials: not with penalty contact (collision_type 1)");
                if (store_F) hipLaunchKernelGGL((k_p2g<R, true, false, true>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, Dc, f);
                else hipLaunchKernelGGL((k_p2g<R, false, false, true>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, Dc, f);
            } else if (pcon) {
                if (store_F) hipLaunchKernelGGL((k_p2g<R, true, true>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, Dc, f);
                else hipLaunchKernelGGL((k_p2g<R, false, true>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, Dc, f);
            } else {
                if (store_F) hipLaunchKernelGGL((k_p2g<R, true, false>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, Dc, f);
                else hipLaunchKernelGGL((k_p2g<R, false, false>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, Dc, f);
            }
            prof_end();
        }
        if (stage != 2) {
            if (fk_in_grid_op) {
                REQUIRE(f + 1 < cfg.max_frames, "forward_kinematics: frame out of range");   // (done by the last workgroup of this substep's k_grid_op launch)
            } else if (!is_recompute && cfg.rigid_velocity_control && D.P > 0 && fk_rides_g2p) {
                REQUIRE(f + 1 < cfg.max_frames, "forward_kinematics: frame out of range");   // (done by the last workgroup of this substep's k_g2p launch)
            } else if (!is_recompute && cfg.rigid_velocity_control && D.P > 0) {  // :329-331, every primitive in one launch
                REQUIRE(f + 1 < cfg.max_frames, "forward_kinematics: frame out of range");
                prof_begin(K_FK);
                hipLaunchKernelGGL(k_prim_fk<double>, dim3(1), dim3(64), 0, stream, D.prim_state, f, D.dt64, D.P, (size_t)cfg.max_frames * 13);
                prof_end();
            }
        }
        Dc.tail_on = 0;
        bool fk_in_contact = false;
        if (tail_done) {
            // no k_grid_op: what rode in its launch moves to the contact kernel's (forward_kinematics to frame f + 1, the emptying of the next hit counter) -
            // or, without a contact launch, to a launch of its own
            const bool contact_launch = stage != 1 && D.any_contact && D.collision_type == CONTACT_MIXED && !D.cloth.present;
            if (fk_in_grid_op && contact_launch) fk_in_contact = true;
            else if (fk_in_grid_op) {
                prof_begin(K_FK);
                hipLaunchKernelGGL(k_prim_fk<double>, dim3(1), dim3(64), 0, stream, D.prim_state, f, D.dt64, D.P, (size_t)cfg.max_frames * 13);
                prof_end();
            }
            if (fuse_next && !contact_launch && D.any_contact) HIP_TRY(hipMemsetAsync(D.nhits_next, 0, sizeof(int), stream));
        } else {
        prof_begin(K_GRID_OP);
        Dc.keep_vmix = stage != 0 ? 1 : 0;                      // the slab phases send v_out - v_mixed across the slab boundaries after the contact pass
        Dc.halo_hs.count = 0;
        if (stage != 0 && halo_in_grid_op && sc.on && halo_buf) {
            Dc.halo_hs = halo_sides(false);
            Dc.halo_np = sc.np;
            Dc.halo_send = halo_buf;
            Dc.halo_recv = halo_buf + 2 * halo_records;
        }
        Dc.zero_next_hits = fuse_next ? 1 : 0;
        Dc.fk_ride = 0;
        if (fk_in_grid_op) { Dc.fk_ride = D.P; Dc.fk_stride = (size_t)cfg.max_frames * 13; }
        if (D.collision_type == CONTACT_GRID && D.any_contact) {
            if (Dc.halo_hs.count) hipLaunchKernelGGL((k_grid_op<R, true, true>), dim3(ngrid_blocks() + (fk_in_grid_op ? 1 : 0)), dim3(BLOCK), 0, stream, Dc, stage);
            else hipLaunchKernelGGL((k_grid_op<R, true>), dim3(ngrid_blocks() + (fk_in_grid_op ? 1 : 0)), dim3(BLOCK), 0, stream, Dc, stage);
        } else if (Dc.halo_hs.count)
            hipLaunchKernelGGL((k_grid_op<R, false, true>), dim3(ngrid_blocks() + (fk_in_grid_op ? 1 : 0)), dim3(BLOCK), 0, stream, Dc, stage);
        else
            hipLaunchKernelGGL((k_grid_op<R, false>), dim3(ngrid_blocks() + (fk_in_grid_op ? 1 : 0)), dim3(BLOCK), 0, stream, Dc, stage);
        Dc.fk_ride = 0;
        prof_end();
        }
        if (stage != 1 && D.any_contact && D.collision_type == CONTACT_MIXED) {
            prof_begin(K_CONTACT);
            Dc.zero_next_hits = 0;
            if (D.cloth.present) {
                hipLaunchKernelGGL(k_cloth_hit_list<R>, dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, Dc, f);
                hipLaunchKernelGGL((k_contact_hits<R, true>), dim3(contact_grad_grid(f)), dim3(BLOCK), 0, stream, Dc, f);
            } else {
                if (tail_done) {
                    Dc.zero_next_hits = fuse_next ? 1 : 0;
                    if (fk_in_contact) { Dc.fk_ride = D.P; Dc.fk_stride = (size_t)cfg.max_frames * 13; }
                }
                hipLaunchKernelGGL((k_contact_hits<R, false>), dim3(contact_grad_grid(f) + (fk_in_contact ? 1 : 0)), dim3(BLOCK), 0, stream, Dc, f);
                Dc.fk_ride = 0; Dc.zero_next_hits = 0;
            }
            prof_end();
        }
        return check_launch();
    }
    // one arena for all frames, sized at the first forward substep from the current active-block count
    bool ck_prepare() {
        if (!ck_enabled) return false;
        if (!ck_arena && !ck_tried) {
            ck_tried = true;
            ck_slot_blocks = (size_t)D.nactive + D.nactive / 4 + 64;
            const size_t bytes = (size_t)cfg.max_frames * ck_slot_blocks * CK_WORDS * sizeof(Vec4<R>);
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes > free_b / 2 ||
                hipMalloc((void**)&ck_arena, bytes) != hipSuccess) {
                (void)hipGetLastError();
                ck_arena = nullptr;
                ck_enabled = false;                        // not enough memory: substep_grad recomputes the forward grid
            }
            if (ck_arena && (hipMalloc((void**)&ck_empty, (size_t)cfg.max_frames * ck_slot_blocks) != hipSuccess ||
                             hipMemsetAsync(ck_empty, 0, (size_t)cfg.max_frames * ck_slot_blocks, stream) != hipSuccess)) {
                (void)hipGetLastError();
                hipFree(ck_empty);
                ck_empty = nullptr;                        // (no flags: every active block is filed)
            }
            // the hit lists ride along when they fit too (otherwise substep_grad repeats the band test).  A slot holds the
            // hits of 1/8 of the particles (the bench has 0.15 % of them inside a contact band); more is reported as an error
            ck_hit_cap = D.Npad / 8 > 8192 ? D.Npad / 8 : 8192;
            if ((size_t)ck_hit_cap > (size_t)D.Npad) ck_hit_cap = D.Npad;
            const size_t hbytes = (size_t)cfg.max_frames * ck_hit_cap * sizeof(Hit);
            if (ck_arena && D.collision_type == CONTACT_MIXED && hipMemGetInfo(&free_b, &total_b) == hipSuccess && hbytes <= free_b / 4) {
                if (hipMalloc((void**)&ck_hits, hbytes) != hipSuccess || hipMalloc((void**)&ck_nhits, cfg.max_frames * sizeof(int)) != hipSuccess) {
                    (void)hipGetLastError();
                    hipFree(ck_hits);
                    ck_hits = nullptr;
                }
            }
        }
        return ck_arena != nullptr && (size_t)D.nactive <= ck_slot_blocks;
    }
    Vec4<R>* ck_slot(int f) { return ck_arena + (size_t)f * ck_slot_blocks * CK_WORDS; }

    // substep :320-337.  phase -1: whole substep; 0 / 1 / 2: the three pieces between which the slab decomposition
    // exchanges halo planes (after 0: {m,p}; after 1: the contact corrections of v_out).
    int substep_phase(int f, const double* action, int phase) {
        int rc;
        ++launch_counter;
        REQUIRE(f >= 0 && f + 1 < cfg.max_frames, "substep: frame f+1 exceeds max_frames");
        REQUIRE(frame_epoch[f] >= 0, "substep: frame f holds no state (call reset/set_frame or simulate up to it first)");
        normalize_grid_set();
        // P2G of this substep already ran inside the G2P launch of the substep before (k_g2p_p2g)?  Anything else than that substep coming next abandons it.
        const bool p2g_done = phase <= 0 && p2g_done_frame == f;       // (phase 0: the slab loop's first piece is where this substep's P2G would run)
        if (!p2g_done && (rc = leave_fused_forward())) return rc;
        p2g_done_frame = -1;
        // the hit counters - and lists - of even and odd frames alternate (k_g2p<R, true> empties the next frame's counter while its save part still reads
        // this one's; k_g2p_p2g appends the next frame's hits while its save part files this frame's list)
        auto bind_hit_counters = [&]() {
            D.nhits = d_nhits + ((f & 1) ? 4 : 0);
            D.nhits_next = d_nhits + ((f & 1) ? 0 : 4);
            D.hits = (f & 1) ? d_hits2 : d_hits;
            D.hits_next = (f & 1) ? d_hits : d_hits2;
        };
        bind_hit_counters();
        bool fuse_next = phase > 0 && fuse_pending_frame == f;           // (decided in this substep's first piece)
        if (phase <= 0) {
            if ((rc = check_contact_supported())) return rc;
            if (action && (rc = set_action(action))) return rc;
            if (D.n_control > 0) {                              // remember what this frame's substep runs with (repair_drift)
                if ((int)action_of_frame.size() < cfg.max_frames) action_of_frame.resize(cfg.max_frames);
                action_of_frame[f] = action_now;
            }
            int e = frame_epoch[f];
            const int repairs_before = drift_repairs;
            if (e == 0 || f - epochs[e].frame >= epochs[e].interval || f < epochs[e].frame) {
                if ((rc = sort_frame(f, e > 0))) return rc;                     // also reports a drift error of the epoch that ends here
                e = frame_epoch[f];
            }
            // A drift repair inside sort_frame replays substeps fs .. f-1 through this function: they re-bound the counter pair to THEIR parity and
            // left counts behind.  Bind for f again and start from two empty counters (the stale binding appended behind nh(f-1) hits and
            // the contact correction of those particles was applied twice).
            bind_hit_counters();
            REQUIRE(!(p2g_done && drift_repairs != repairs_before), "internal: a re-sort in front of a substep whose P2G has run");
            if (!p2g_done && (nhits_zero_frame != f || drift_repairs != repairs_before)) {
                HIP_TRY(hipMemsetAsync(d_nhits, 0, 2 * sizeof(int), stream));
                HIP_TRY(hipMemsetAsync(d_nhits + 4, 0, sizeof(int), stream));
            }
            nhits_zero_frame = -1;
            if ((rc = bind_epoch(e))) return rc;
            ck_epoch[f] = -1;
            // Fused forward step (k_g2p_p2g): the batched loop (smac_substeps) has announced that substep f + 1 follows, it keeps this binning, and nothing
            // it needs changes in between (no particle action, penalty contact, sheet or second material: those P2G instantiations stay on their own) -
            // then this substep's G2P launch also runs the next substep's P2G, and the call for f + 1 starts at k_grid_op.  SMAC_FUSED_FWD=0: off.
            // (the slab loop's pieces take it too: G2P of substep f comes after the contact exchange of f, P2G of f + 1 before the {m,p} exchange of f + 1)
            fuse_next = fused_fwd_env && fwd_hint == f + 1 && f + 2 < cfg.max_frames && !repairing && D.nchunks > 0 && D.nactive > 0 &&
                        f + 1 - epochs[e].frame < epochs[e].interval && D.n_control == 0 && D.collision_type != CONTACT_PARTICLE && !D.cloth.present && !D.mat_id;
            fuse_pending_frame = fuse_next && phase == 0 ? f : -1;
            // whole substep with particles: forward_kinematics rides in k_g2p's launch instead of a launch of its own (SMAC_FK_RIDE=0: own kernel)
            fk_rides_g2p = !fuse_next && fk_ride_env && phase < 0 && cfg.rigid_velocity_control && D.P > 0 && D.nchunks > 0 && D.nactive > 0;
            if ((rc = forward_grid(f, true, false, phase < 0 ? 0 : 1, p2g_done, fuse_next))) return rc;
        }
        if (phase == 1 && (rc = forward_grid(f, true, false, 2))) return rc;
        if (phase < 0 || phase == 2) {
            const int e = frame_epoch[f];
            const bool save = ck_epoch[f] != e && D.nchunks > 0 && D.n_control == 0 && ck_prepare();   // keep the forward grid for substep_grad
            const bool keep_hits = save && ck_hits && any_contact();
            const bool save_in_g2p = save && save_in_g2p_env;                              // the save rides in k_g2p's launch (SMAC_SAVE_IN_G2P=0: own kernel)
            if (save) {
                ck_has_hits[f] = keep_hits ? 1 : 0;
                h_nhits[f] = -1;                                   // (unknown until the saving launch has run)
                if (!save_in_g2p) {
                    prof_begin(K_CKPT);
                    DevSim<R> Ds = D;
                    Ds.ck_flags = ck_flags_of(f);
                    Ds.save_nhits_host = keep_hits ? h_nhits + f : nullptr;
                    hipLaunchKernelGGL(k_grid_save<R>, dim3(ngrid_blocks()), dim3(BLOCK), 0, stream, Ds, ck_slot(f),
                                       keep_hits ? ck_hits + (size_t)f * ck_hit_cap : (Hit*)nullptr, keep_hits ? ck_nhits + f : (int*)nullptr, ck_hit_cap);
                    prof_end();
                }
                ck_epoch[f] = e;
                ck_gen[f] = config_gen;
            }
            if (D.nchunks > 0) {
                prof_begin(fuse_next ? K_G2P_P2G : K_G2P);
                D.check_next = (f + 1 - epochs[e].frame < epochs[e].interval) ? 1 : 0;     // else substep(f+1) re-bins first
                if (fuse_next) {                                                            // G2P of this substep + P2G of the next in one launch
                    DevSim<R> Dg = D;
                    Dg.any_contact = any_contact() ? 1 : 0;
                    // tail reduction of the riding P2G (whole substeps only: the slab pieces exchange {m,p} between P2G and grid_op).  The save part of this
                    // launch still reads this substep's {m,p} / v_out: its waves arrive too (tail_extra)
                    Dg.tail_on = (tail_env && phase < 0 && !(D.collision_type == CONTACT_GRID && Dg.any_contact)) ? 1 : 0;
                    Dg.tail_extra = (Dg.tail_on && save_in_g2p) ? 1 : 0;
                    p2g_tail_frame = Dg.tail_on ? f + 1 : -1;
                    if (save_in_g2p) {
                        Dg.save_ck = ck_slot(f);
                        Dg.save_hits = keep_hits ? ck_hits + (size_t)f * ck_hit_cap : (Hit*)nullptr;
                        Dg.save_nhits = keep_hits ? ck_nhits + f : (int*)nullptr;
                        Dg.save_nhits_host = keep_hits ? h_nhits + f : nullptr;
                        Dg.save_hit_cap = ck_hit_cap;
                        Dg.ck_flags = ck_flags_of(f);
                        Dg.save_blocks = (ngrid_blocks() + 7) & ~7;
                        hipLaunchKernelGGL((k_g2p_p2g<R, true>), dim3(Dg.save_blocks + nchunk_blocks()), dim3(BLOCK), 0, stream, Dg, f);
                    } else
                        hipLaunchKernelGGL((k_g2p_p2g<R, false>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, Dg, f);
                    p2g_done_frame = f + 1;
                } else if (save_in_g2p) {
                    DevSim<R> Dg = D;
                    if (fk_rides_g2p) { Dg.fk_ride = D.P; Dg.fk_stride = (size_t)cfg.max_frames * 13; }
                    Dg.save_ck = ck_slot(f);
                    Dg.save_hits = keep_hits ? ck_hits + (size_t)f * ck_hit_cap : (Hit*)nullptr;
                    Dg.save_nhits = keep_hits ? ck_nhits + f : (int*)nullptr;
                    Dg.save_nhits_host = keep_hits ? h_nhits + f : nullptr;
                    Dg.save_hit_cap = ck_hit_cap;
                    Dg.ck_flags = ck_flags_of(f);
                    Dg.save_blocks = (ngrid_blocks() + 7) & ~7;                              // (a multiple of 8: the chunk -> XCD dealing of the g2p part stays aligned)
                    hipLaunchKernelGGL((k_g2p<R, true>), dim3(Dg.save_blocks + nchunk_blocks() + (fk_rides_g2p ? 1 : 0)), dim3(BLOCK), 0, stream, Dg, f);
                } else {
                    DevSim<R> Dg = D;
                    if (fk_rides_g2p) { Dg.fk_ride = D.P; Dg.fk_stride = (size_t)cfg.max_frames * 13; }
                    hipLaunchKernelGGL((k_g2p<R, false>), dim3(nchunk_blocks() + (fk_rides_g2p ? 1 : 0)), dim3(BLOCK), 0, stream, Dg, f);
                }
                fk_rides_g2p = false;
                nhits_zero_frame = f + 1;                                                    // every form of k_g2p leaves the next frame's hit counter empty
                prof_end();
            }
            frame_epoch[f + 1] = e;
            if (!repairing && f + 1 < (int)cloth_ops_of_frame.size()) cloth_ops_of_frame[f + 1].clear();   // a new frame f + 1: the host's calls on it start over
            fwd_head = f + 1;
            bwd_since_fwd = false;
            g2p_done_frame = -1;
        }
        if (phase >= 0) slab_phase_used = true;
        return check_launch();
    }
    int substep(int f, const double* action) override { return substep_phase(f, action, -1); }

    // substep_grad :339-378.  phase -1: whole; 0: restore/recompute + g2p.grad (+ slab reduction of grid_v_out.grad);
    // 1: contact adjoint; 2: grid_op.grad, kinematics adjoint, p2g.grad.  Halo sums of grid_v_out.grad follow phase 0,
    // of grid_v_mixed.grad phase 1.
    bool pending_adj_zero = false;
    // Fused backward step (k_p2g_g2p_grad): when the batched loop (smac_substeps_grad) announces that substep f - 1 is reversed next, the
    // p2g.grad launch of substep f also does the G2P adjoint of substep f - 1 - its forward grid is restored first - and the call for f - 1
    // resumes at the slab reduction.  SMAC_FUSED_PG=0 keeps the two kernels apart.
    int fused_pg_env = getenv("SMAC_FUSED_PG") ? atoi(getenv("SMAC_FUSED_PG")) : 1;
    int save_in_g2p_env = getenv("SMAC_SAVE_IN_G2P") ? atoi(getenv("SMAC_SAVE_IN_G2P")) : 1;
    int fk_ride_env = getenv("SMAC_FK_RIDE") ? atoi(getenv("SMAC_FK_RIDE")) : 1;   // forward_kinematics inside k_g2p's launch, its adjoint inside the grid-adjoint reduction's
    bool fk_rides_g2p = false;
    bool fk_grad_rode = false;           // this substep_grad call's forward_kinematics.grad already ran inside its reduction launch
    // Restore-ahead (k_reduce_grid_grad_ahead): inside the fused batched sweep the forward grid of substep f - 1 is restored by the launch that
    // reduces substep f's grid adjoint, into the second of two buffer sets {grid_in, grid_mixed, grid_out, grid_out.grad}; the sets change roles after
    // the fused particle kernel.  Outside the sweep everything lives in set 0 (normalize_grid_set).  SMAC_RESTORE_AHEAD=0: k_grid_restore as before.
    int restore_ahead_env = getenv("SMAC_RESTORE_AHEAD") ? atoi(getenv("SMAC_RESTORE_AHEAD")) : 1;
    Vec4<R>* grid_alt = nullptr;         // set 1: 4 fields
    bool grid_alt_tried = false;
    int grid_set = 0;                    // the set D.vin / vmix / vout / aout point to
    int ahead_frame = -1;                // frame whose forward grid the last reduction launch restored into the OTHER set
    int hits_in_place_frame = -1;        // frame whose contact adjoint walks the filed hit list in place (its restore did not copy it)
    GridSet<R> grid_set_ptrs(int which) const {
        if (which == 0) return GridSet<R>{grid_block, grid_block + D.G, grid_block + 2 * D.G, grid_block + 5 * D.G};
        return GridSet<R>{grid_alt, grid_alt + D.G, grid_alt + 2 * D.G, grid_alt + 3 * D.G};
    }
    void use_grid_set(int which) {
        const GridSet<R> g = grid_set_ptrs(which);
        D.vin = g.vin; D.vmix = g.vmix; D.vout = g.vout; D.aout = g.aout;
        grid_set = which;
    }
    bool grid_alt_ready() {
        if (!grid_alt && !grid_alt_tried) {
            grid_alt_tried = true;
            size_t free_b = 0, total_b = 0;
            const size_t bytes = 4 * D.G * sizeof(Vec4<R>);
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes > free_b / 4 || hipMalloc((void**)&grid_alt, bytes) != hipSuccess) {
                (void)hipGetLastError();
                grid_alt = nullptr;
            } else if (hipMemsetAsync(grid_alt, 0, bytes, stream) != hipSuccess) {
                hipFree(grid_alt);
                grid_alt = nullptr;
            }
        }
        return grid_alt != nullptr;
    }
    // back to set 0 (whose contents are then stale: the flags say so).  Called wherever the fused sweep is not being continued.
    void normalize_grid_set() {
        ahead_frame = -1;
        hits_in_place_frame = -1;
        if (grid_set != 0) {
            use_grid_set(0);
        }
    }
    int nhits_zero_frame = -1;           // forward frame whose hit counter is known to be empty (k_g2p<R, true> of the frame before emptied it)
    int fused_fwd_env = getenv("SMAC_FUSED_FWD") ? atoi(getenv("SMAC_FUSED_FWD")) : 1;
    int fwd_hint = -1;                   // substep the caller runs next (-1: unknown; smac_substeps announces it)
    int p2g_done_frame = -1;             // substep whose P2G already ran inside the G2P launch of the substep before it
    int fuse_pending_frame = -1;         // slab pieces: the substep whose last piece (G2P) will carry the next substep's P2G
    bool halo_in_grid_op = false;        // set by smac_substeps_slab around the two k_grid_op pieces of a substep (DevSim::halo_hs)
    int halo_fuse_env = getenv("SMAC_HALO_FUSE") ? atoi(getenv("SMAC_HALO_FUSE")) : 1;
    void hint_forward_next(int f) override { fwd_hint = f; }
    // A P2G that ran ahead (k_g2p_p2g) and whose substep is not the next thing to happen: what it left in the slabs is overwritten by the next P2G, what it
    // added to D.vdrift and to the hit counter is taken back here.  (Only an error between two substeps of smac_substeps gets here.)
    int leave_fused_forward() {
        if (p2g_done_frame < 0) return SMAC_OK;
        p2g_done_frame = -1;
        p2g_tail_frame = -1;
        nhits_zero_frame = -1;
        HIP_TRY(hipMemsetAsync(vdrift, 0, D.G * sizeof(Vec4<R>), stream));
        HIP_TRY(hipMemsetAsync(d_tail_cnt, 0, ((size_t)nblocks + 1) * sizeof(int), stream));
        HIP_TRY(hipMemsetAsync(d_nhits, 0, 2 * sizeof(int), stream));
        HIP_TRY(hipMemsetAsync(d_nhits + 4, 0, sizeof(int), stream));
        return SMAC_OK;
    }
    int bwd_hint = -1;                   // frame the caller will reverse next (-1: unknown)
    int g2p_done_frame = -1;             // substep whose restore + g2p.grad already ran inside the previous call
    bool g2p_done_paz = false;           // ... and whether its adjoint frame started from zero
    void hint_backward_next(int f) override { bwd_hint = f; }
    bool can_fuse_prev(int f, int e, int phase, const double* action_grad_out) {
        // (phase 2: the slab loop's last piece - p2g.grad(f) and g2p.grad(f - 1) cross no exchange; its grid adjoint pass stays the three-kernel sequence)
        if (!fused_pg_env || sizeof(R) != 4 || (phase >= 0 && phase != 2) || bwd_hint != f - 1 || f < 1 || action_grad_out) return false;   // (f64: 256 VGPRs + 88 KB of LDS, one workgroup per CU)
        if (!pending_adj_zero || rolling()) return false;                                                   // frame f carried a seed / frames come and go
        if (phase < 0 ? !fused_grid_bwd(phase) : (D.collision_type == CONTACT_GRID && any_contact())) return false;
        if (frame_epoch[f - 1] != e || !(adj_epoch[f - 1] < 0 || adj_epoch[f - 1] == e)) return false;        // a re-sort lies between the two substeps
        if (!(ck_arena && ck_epoch[f - 1] == e && ck_gen[f - 1] == config_gen && D.n_control == 0 && D.nchunks > 0)) return false;
        if (D.collision_type == CONTACT_PARTICLE || D.cloth.present || D.mat_id) return false;
        if (any_contact() && D.collision_type == CONTACT_MIXED && !ck_has_hits[f - 1]) return false;          // (the band test would have to run in between)
        return true;
    }
    // whole-substep backward without grid-node contact: k_reduce_grid_grad does k_reduce_aout's and k_grid_op_grad's work in one pass
    // (SMAC_FUSED_GRID_BWD=0 keeps the three-kernel sequence the slab phases use)
    int fused_bwd_env = getenv("SMAC_FUSED_GRID_BWD") ? atoi(getenv("SMAC_FUSED_GRID_BWD")) : 1;
    bool fused_grid_bwd(int phase) const {
        return fused_bwd_env && phase < 0 && !(D.collision_type == CONTACT_GRID && any_contact());
    }
    int substep_grad_phase(int f, const double* action, const double* ext_f_grad, double* action_grad_out, int phase) {
        int rc;
        ++launch_counter;
        if ((rc = need_grad()) || (rc = leave_fused_forward())) return rc;
        D.nhits = d_nhits;                   // (the backward pass uses one counter; the forward pass re-binds and re-empties its pair)
        nhits_zero_frame = -1;
        if (phase <= 0) fk_grad_rode = false;
        if (phase <= 0 && !bwd_since_fwd) {                 // first backward substep after a forward pass: a drifted epoch is repaired (or reported) now
            if ((rc = check_drift())) return rc;
            bwd_since_fwd = true;
        }
        REQUIRE(f >= 0 && f + 1 < cfg.max_frames, "substep_grad: frame f+1 exceeds max_frames");
        REQUIRE(frame_epoch[f] > 0, "substep_grad: frame f was not produced/consumed by a forward substep");
        const int e = frame_epoch[f];
        if (g2p_done_frame >= 0 && (g2p_done_frame != f || phase > 0)) {
            g2p_done_frame = -1;
            normalize_grid_set();
            REQUIRE(false, "substep_grad: the batched backward sweep was interrupted (its next substep had been started)");
        }
        if (!(phase <= 0 && g2p_done_frame == f)) normalize_grid_set();
        if (phase == 0 && g2p_done_frame == f) {          // the slab loop's first piece: restore + g2p.grad ran inside the previous substep's last piece
            g2p_done_frame = -1;
            if ((rc = stage_ext_f_grad(ext_f_grad))) return rc;
            D.Af = adj_ptr(f);
            D.An = adj_ptr(f + 1);
            D.An_map = nullptr;
            pending_adj_zero = g2p_done_paz;
            D.cur_frame = f;
            prof_begin(K_REDUCE);
            if (halo_in_bwd && sc.on && halo_buf) hipLaunchKernelGGL((k_reduce_aout<R, true>), dim3(ngrid_blocks()), dim3(BLOCK), 0, stream, with_halo(D));
            else hipLaunchKernelGGL((k_reduce_aout<R, false>), dim3(ngrid_blocks()), dim3(BLOCK), 0, stream, D);
            prof_end();
        } else if (phase < 0 && g2p_done_frame == f) {    // restore + g2p.grad of this substep ran inside the previous call (k_p2g_g2p_grad)
            g2p_done_frame = -1;
            if ((rc = stage_ext_f_grad(ext_f_grad))) return rc;
            D.Af = adj_ptr(f);
            D.An = adj_ptr(f + 1);                          // same epoch (can_fuse_prev): in this order already
            D.An_map = nullptr;
            pending_adj_zero = g2p_done_paz;
            D.cur_frame = f;
            const bool tail_did = bwd_tail_frame == f;        // tail reduction: the fused launch of the call before also finished this substep's grid_v_out.grad / grid_v_in.grad
            bwd_tail_frame = -1;
            DevSim<R> Dr = D;
            fk_grad_rode = fk_ride_env && cfg.rigid_velocity_control && D.P > 0;     // forward_kinematics.grad of this substep: the last D.P workgroups of the launch
            if (fk_grad_rode) { Dr.fk_ride = D.P; Dr.fk_stride = (size_t)cfg.max_frames * 13; }
            Dr.ck_flags_next = f > 0 ? ck_flags_of(f - 1) : nullptr;
            Dr.ck_flags = reduce_flags(f, e);
            const bool ahead_ok = restore_ahead_env && can_fuse_prev(f, e, phase, action_grad_out) && grid_alt_ready();
            if (tail_did) {
                // no reduction launch: the restore of substep f - 1's forward grid into the other buffer set and forward_kinematics.grad ride in the contact
                // adjoint's launch below - or, for a frame without hits, in a launch of their own (k_restore_ahead)
                pend_restore = ahead_ok;
                pend_fk = fk_grad_rode;
                if (ahead_ok) ahead_frame = f - 1;
            } else {
            prof_begin(K_REDUCE);
            if (ahead_ok) {
                // this substep will hand over to substep f - 1 inside k_p2g_g2p_grad: its forward grid is restored by THIS launch, into the other set
                hipLaunchKernelGGL(k_reduce_grid_grad_ahead<R>, dim3(2 * ngrid_blocks() + Dr.fk_ride), dim3(BLOCK), 0, stream, Dr, grid_set_ptrs(1 - grid_set),
                                   (const Vec4<R>*)ck_slot(f - 1));
                ahead_frame = f - 1;
            } else
                hipLaunchKernelGGL(k_reduce_grid_grad<R>, dim3(ngrid_blocks() + Dr.fk_ride), dim3(BLOCK), 0, stream, Dr);
            prof_end();
            }
        } else if (phase <= 0) {
            hits_from_ck_frame = -1;
            if ((rc = check_contact_supported())) return rc;
            if (action && (rc = set_action(action))) return rc;
            if ((rc = stage_ext_f_grad(ext_f_grad))) return rc;                   // :342-344
            if ((rc = bind_epoch(e))) return rc;
            // adjoint of frame f+1 in this epoch's particle order; adjoint of frame f must be in it too
            const R* An = nullptr;
            if (adj_epoch[f + 1] < 0 && (rc = adj_make_zero(f + 1))) return rc;  // no seed and no later substep: zero adjoint
            const int* An_map = nullptr;
            bool direct = false;
            if ((rc = adjoint_direct_map(f + 1, e, &An_map, &direct))) return rc;
            if (direct) An = adj_ptr(f + 1);                                      // read in place, through the map where the binning differs
            else if ((rc = adjoint_in_order(f + 1, e, &An))) return rc;
            REQUIRE(An, kPoolMessage);
            if (adj_epoch[f] >= 0 && adj_epoch[f] != e) {                         // seeds stored in another order: convert in place
                const R* tmp = nullptr;
                R* dst = tmp_frame;
                if (An == tmp_frame) {                                            // frame f+1 already sits there: use the second scratch frame
                    if (!tmp_frame2) HIP_TRY(hipMalloc((void**)&tmp_frame2, frame_scalars() * sizeof(R)));
                    dst = tmp_frame2;
                }
                if ((rc = adjoint_in_order(f, e, &tmp, dst))) return rc;
                HIP_TRY(hipMemcpyAsync(adj_ptr(f), tmp, frame_scalars() * sizeof(R), hipMemcpyDeviceToDevice, stream));
            }
            D.Af = adj_ptr(f);
            REQUIRE(D.Af, kPoolMessage);
            pending_adj_zero = adj_epoch[f] < 0;                                  // A[f] is known to be all zero: write instead of +=
            adj_epoch[f] = e;
            adj_stale[f] = 0;                                                     // write mode overwrites every row
            D.An = An;
            D.An_map = An_map;
            const bool ck_ok = ck_arena && ck_epoch[f] == e && ck_gen[f] == config_gen && D.n_control == 0 && D.nchunks > 0;
            if (ck_ok) {
                // forward grid of this frame is on file: restore it (and zero the grid adjoints) instead of recomputing
                D.any_contact = any_contact() ? 1 : 0;
                D.cur_frame = f;
                prof_begin(K_CKPT);
                const bool have_hits = ck_has_hits[f] && D.any_contact && D.collision_type == CONTACT_MIXED;
                DevSim<R> Dk = D;
                Dk.ck_flags = ck_flags_of(f);
                hipLaunchKernelGGL(k_grid_restore<R>, dim3(ngrid_blocks()), dim3(BLOCK), 0, stream, Dk, (const Vec4<R>*)ck_slot(f),
                                   have_hits ? (const Hit*)(ck_hits + (size_t)f * ck_hit_cap) : (const Hit*)nullptr,
                                   have_hits ? (const int*)(ck_nhits + f) : (const int*)nullptr, fused_grid_bwd(phase) ? 0 : 1);
                if (have_hits) hits_from_ck_frame = f;
                if (D.any_contact && D.collision_type != CONTACT_GRID && !have_hits) {
                    if (D.cloth.present) hipLaunchKernelGGL(k_cloth_hit_list<R>, dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, D, f);
                    else hipLaunchKernelGGL(k_contact_mask<R>, dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, D, f);
                }
                prof_end();
            } else {
                REQUIRE(phase < 0, "slab-decomposed substep_grad needs the forward-grid checkpoint of this frame "
                                   "(recomputing it would need the forward halo exchanges again)");
                if ((rc = forward_grid(f, false, true))) return rc;               // :347-359 (clears values + adjoints, recomputes)
            }
            if (D.nchunks > 0) {
                prof_begin(K_G2P_GRAD);
                if (pending_adj_zero) hipLaunchKernelGGL((k_g2p_grad<R, false>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, D, f);   // :361
                else hipLaunchKernelGGL((k_g2p_grad<R, true>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, D, f);
                prof_end();
                prof_begin(K_REDUCE);
                if (fused_grid_bwd(phase)) {
                    DevSim<R> Dr = D;
                    fk_grad_rode = fk_ride_env && cfg.rigid_velocity_control && D.P > 0;
                    if (fk_grad_rode) { Dr.fk_ride = D.P; Dr.fk_stride = (size_t)cfg.max_frames * 13; }
                    Dr.ck_flags = ck_ok ? reduce_flags(f, e) : nullptr;      // (a recomputed forward grid has no flags)
                    hipLaunchKernelGGL(k_reduce_grid_grad<R>, dim3(ngrid_blocks() + Dr.fk_ride), dim3(BLOCK), 0, stream, Dr);
                } else if (halo_in_bwd && phase == 0 && sc.on && halo_buf) hipLaunchKernelGGL((k_reduce_aout<R, true>), dim3(ngrid_blocks()), dim3(BLOCK), 0, stream, with_halo(D));
                else hipLaunchKernelGGL((k_reduce_aout<R, false>), dim3(ngrid_blocks()), dim3(BLOCK), 0, stream, D);
                prof_end();
            }
        }
        if (hits_in_place_frame == f) hits_from_ck_frame = f;
        const bool no_hits = hits_from_ck_frame == f && h_nhits[f] == 0;        // the frame's filed hit list is empty: nothing to reverse
        if (no_hits && (phase < 0 || phase == 1)) ++contact_skips;
        if ((phase < 0 || phase == 1) && D.nchunks > 0 && D.collision_type == CONTACT_MIXED && any_contact() && !no_hits) {   // :362-363, 389-393
            prof_begin(K_CONTACT_GRAD);
            DevSim<R> Dc = D;
            if (hits_in_place_frame == f) {                // (restored ahead: the hit list was not copied out of the checkpoint)
                Dc.hits = ck_hits + (size_t)f * ck_hit_cap;
                Dc.nhits = ck_nhits + f;
            }
            const GridSet<R> none{nullptr, nullptr, nullptr, nullptr};
            const int cgrid = contact_grad_grid(f, hits_from_ck_frame == f);          // (the filed list's own count when the list is the filed one)
            if (D.cloth.present) {
                if (fused_grid_bwd(phase)) hipLaunchKernelGGL((k_contact_grad<R, true, true>), dim3(cgrid), dim3(BLOCK), 0, stream, Dc, f, none, (const Vec4<R>*)nullptr, 0);
                else hipLaunchKernelGGL((k_contact_grad<R, false, true>), dim3(cgrid), dim3(BLOCK), 0, stream, Dc, f, none, (const Vec4<R>*)nullptr, 0);
            } else if (fused_grid_bwd(phase)) {
                // (tail reduction) what the reduction launch used to carry rides here: the next frame's restore in the first workgroups, forward_kinematics.grad in the last
                const int ride = pend_restore ? ngrid_blocks() : 0, fkr = pend_fk ? D.P : 0;
                Dc.fk_ride = fkr; Dc.fk_stride = (size_t)cfg.max_frames * 13;
                Dc.ck_flags_next = f > 0 ? ck_flags_of(f - 1) : nullptr;
                hipLaunchKernelGGL((k_contact_grad<R, true, false>), dim3(ride + cgrid + fkr), dim3(BLOCK), 0, stream, Dc, f,
                                   pend_restore ? grid_set_ptrs(1 - grid_set) : none, pend_restore ? (const Vec4<R>*)ck_slot(f - 1) : (const Vec4<R>*)nullptr, ride);
                pend_restore = pend_fk = false;
            } else hipLaunchKernelGGL((k_contact_grad<R, false, false>), dim3(cgrid), dim3(BLOCK), 0, stream, Dc, f, none, (const Vec4<R>*)nullptr, 0);
            prof_end();
        }
        if (pend_restore || pend_fk) {                       // (tail reduction, no contact adjoint launch in this substep)
            prof_begin(K_REDUCE);
            DevSim<R> Dr = D;
            Dr.fk_ride = pend_fk ? D.P : 0; Dr.fk_stride = (size_t)cfg.max_frames * 13;
            Dr.ck_flags_next = f > 0 ? ck_flags_of(f - 1) : nullptr;
            hipLaunchKernelGGL(k_restore_ahead<R>, dim3((pend_restore ? ngrid_blocks() : 0) + Dr.fk_ride), dim3(BLOCK), 0, stream, Dr,
                               pend_restore ? grid_set_ptrs(1 - grid_set) : GridSet<R>{nullptr, nullptr, nullptr, nullptr},
                               pend_restore ? (const Vec4<R>*)ck_slot(f - 1) : (const Vec4<R>*)nullptr, pend_restore ? 1 : 0);
            prof_end();
            pend_restore = pend_fk = false;
        }
        if (phase < 0 || phase == 2) {
            if (D.nchunks > 0 && !fused_grid_bwd(phase)) {
                prof_begin(K_GRID_OP_GRAD);
                if (D.collision_type == CONTACT_GRID && any_contact())
                    hipLaunchKernelGGL((k_grid_op_grad<R, true>), dim3(ngrid_blocks()), dim3(BLOCK), 0, stream, D);   // :394 / :365
                else if (halo_in_bwd && phase == 2 && sc.on && halo_buf)
                    hipLaunchKernelGGL((k_grid_op_grad<R, false, true>), dim3(ngrid_blocks()), dim3(BLOCK), 0, stream, with_halo(D));
                else
                    hipLaunchKernelGGL((k_grid_op_grad<R, false>), dim3(ngrid_blocks()), dim3(BLOCK), 0, stream, D);
                prof_end();
            }
            if (cfg.rigid_velocity_control && D.P > 0 && !fk_grad_rode) {         // :367-369 (the primitives do not interact: one launch)
                prof_begin(K_FK);
                hipLaunchKernelGGL(k_prim_fk_grad<double>, dim3(D.P), dim3(64), 0, stream, (const double*)D.prim_state, D.prim_grad, f, D.dt64,
                                   (size_t)cfg.max_frames * 13);
                prof_end();
            }
            if (D.nchunks > 0 && can_fuse_prev(f, e, phase, action_grad_out)) {
                R* Af_prev = adj_ptr(f - 1);
                REQUIRE(Af_prev, kPoolMessage);
                const bool paz_prev = adj_epoch[f - 1] < 0;
                const bool have_hits = ck_has_hits[f - 1] && D.any_contact && D.collision_type == CONTACT_MIXED;
                const bool ahead = ahead_frame == f - 1;
                if (ahead) {
                    use_grid_set(1 - grid_set);            // the reduction launch of this call restored frame f - 1 there; nothing below reads frame f's grid
                    hits_in_place_frame = have_hits ? f - 1 : -1;
                    ahead_frame = -1;
                } else {
                    prof_begin(K_CKPT);
                    DevSim<R> Dk = D;
                    Dk.ck_flags = ck_flags_of(f - 1);
                    hipLaunchKernelGGL(k_grid_restore<R>, dim3(ngrid_blocks()), dim3(BLOCK), 0, stream, Dk, (const Vec4<R>*)ck_slot(f - 1),
                                       have_hits ? (const Hit*)(ck_hits + (size_t)(f - 1) * ck_hit_cap) : (const Hit*)nullptr,
                                       have_hits ? (const int*)(ck_nhits + (f - 1)) : (const int*)nullptr, phase < 0 ? 0 : 2);
                    prof_end();
                    hits_in_place_frame = -1;
                    hits_from_ck_frame = have_hits ? f - 1 : -1;
                }
                DevSim<R> D2 = D;
                D2.Af_prev = Af_prev;
                // tail reduction: the launch also completes substep f - 1's grid_v_out.grad and pushes it through grid_op's node adjoint (whole substeps; the slab
                // pieces exchange grid_v_out.grad between the scatter and the node adjoint)
                D2.tail_on = (tail_env && tail_bwd_env && phase < 0) ? 1 : 0;
                D2.tail_extra = 0;
                D2.ck_flags = D2.tail_on ? reduce_flags(f - 1, e) : nullptr;
                bwd_tail_frame = D2.tail_on ? f - 1 : -1;
                prof_begin(K_P2G_G2P_GRAD);
                if (paz_prev) hipLaunchKernelGGL((k_p2g_g2p_grad<R, false>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, D2, f);
                else hipLaunchKernelGGL((k_p2g_g2p_grad<R, true>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, D2, f);
                prof_end();
                adj_epoch[f - 1] = e;
                adj_stale[f - 1] = 0;
                g2p_done_frame = f - 1;
                g2p_done_paz = paz_prev;
            } else if (D.nchunks > 0) {
                prof_begin(K_P2G_GRAD);
                if (D.mat_id) {
                    if (pending_adj_zero) hipLaunchKernelGGL((k_p2g_grad<R, false, true, true>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, D, f);
                    else hipLaunchKernelGGL((k_p2g_grad<R, true, true, true>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, D, f);
                } else if (pending_adj_zero) hipLaunchKernelGGL((k_p2g_grad<R, false, true>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, D, f);   // :371-374
                else hipLaunchKernelGGL((k_p2g_grad<R, true, true>), dim3(nchunk_blocks()), dim3(BLOCK), 0, stream, D, f);
                prof_end();
                if (D.collision_type == CONTACT_PARTICLE && any_contact()) {      // adjoint of p2g's contact impulse (:203-206)
                    prof_begin(K_CONTACT_GRAD);
                    if (D.cloth.present) hipLaunchKernelGGL((k_particle_contact_grad<R, true>), dim3(256), dim3(BLOCK), 0, stream, D, f);
                    else hipLaunchKernelGGL((k_particle_contact_grad<R, false>), dim3(256), dim3(BLOCK), 0, stream, D, f);
                    prof_end();
                }
            }
            if ((rc = check_launch())) return rc;
            if (action_grad_out && D.n_control > 0) {                             // :378
                R tmp[3 * 64];
                HIP_TRY(hipMemcpyAsync(tmp, D.action_grad, 3 * D.n_control * sizeof(R), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                for (int i = 0; i < 3 * D.n_control; ++i) action_grad_out[i] = (double)tmp[i];
            }
            adj_release(f + 2);                    // rolling storage: the sweep has passed frame f+2 two substeps ago
            if (g2p_done_frame < 0) normalize_grid_set();   // the sweep does not continue inside this epoch: back to buffer set 0
        }
        return check_launch();
    }
    int substep_grad(int f, const double* action, const double* ext_f_grad, double* action_grad_out) override {
        return substep_grad_phase(f, action, ext_f_grad, action_grad_out, -1);
    }
    int substep_phase_v(int f, int phase) override { return substep_phase(f, nullptr, phase); }
    int substep_grad_phase_v(int f, const double* ext_f_grad, int phase) override { return substep_grad_phase(f, nullptr, ext_f_grad, nullptr, phase); }

    // ---- halo planes (slab decomposition) -------------------------------------------------------
    Vec4<R>* field_by_name(const char* name) {
        struct { const char* n; Vec4<R>* p; } tab[] = {{"grid_in", D.vin}, {"grid_mixed", D.vmix}, {"grid_out", D.vout},
                                                       {"grid_in.grad", D.ain}, {"grid_mixed.grad", D.amix}, {"grid_out.grad", D.aout}};
        for (auto& t : tab)
            if (!strcmp(t.n, name)) return t.p;
        return nullptr;
    }
    int halo_pack(const char* field, int plane0, int np, void* dev_out, int minus_mixed) override {
        Vec4<R>* fp = field ? field_by_name(field) : nullptr;
        REQUIRE(fp && dev_out && plane0 >= 0 && np > 0 && plane0 + np <= D.n, "halo_pack: bad field / plane range");
        REQUIRE(grid_epoch > 0, "halo_pack: no epoch bound");
        const int total = np * D.n * D.n;
        hipLaunchKernelGGL(k_halo_pack<R>, dim3(nblk(total)), dim3(BLOCK), 0, stream, D, (const Vec4<R>*)fp,
                           (const Vec4<R>*)(minus_mixed ? D.vmix : nullptr), plane0, np, (Vec4<R>*)dev_out);
        return check_launch();
    }
    int halo_unpack_add(const char* field, int plane0, int np, const void* dev_in) override {
        Vec4<R>* fp = field ? field_by_name(field) : nullptr;
        REQUIRE(fp && dev_in && plane0 >= 0 && np > 0 && plane0 + np <= D.n, "halo_unpack_add: bad field / plane range");
        REQUIRE(grid_epoch > 0, "halo_unpack_add: no epoch bound");
        const int total = np * D.n * D.n;
        hipLaunchKernelGGL(k_halo_unpack_add<R>, dim3(nblk(total)), dim3(BLOCK), 0, stream, D, fp, plane0, np, (const Vec4<R>*)dev_in);
        return check_launch();
    }

    // ---- slab decomposition INSIDE the library: per substep the neighbour-only exchanges of the shared x-planes run on RCCL
    // (ncclSend / ncclRecv in one group per exchange) on their own stream, ordered against the kernels' stream with events - no host
    // round trip, no Python between the phases.  parallel.SlabRunner (torch.distributed) remains as the logic oracle of the CPU tests.
    ncclComm_t comm = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_kernels = nullptr, ev_comm = nullptr;
    int c_rank = 0, c_world = 1;
    struct SlabCfg { bool on = false; int left0 = 0, right0 = 0, np = 2, peer_l = -1, peer_r = -1; bool contact_l = false, contact_r = false, self_loop = false; } sc;
    Vec4<R>* halo_buf = nullptr;        // [send L | send R | recv L | recv R], np * n * n records each
    size_t halo_records = 0;
    // 1: pack / events / unpack without the RCCL calls (world-1 self loop: host-enqueue measurements on one GPU).  2: the IPC link of smac_comm.hpp -
    // DIFFERENT ranks that may share one GPU, messages through exported device mailboxes, host-synchronous (the test transport of the distinct-peer code)
    int comm_stub = getenv("SMAC_COMM_STUB") ? atoi(getenv("SMAC_COMM_STUB")) : 0;
    IpcLink ipc;
    // both directions of one neighbour exchange over the IPC link: send[s] / recv[s], s = 0 left, 1 right; a side takes part when either message is non-empty
    int ipc_sendrecv(const void* const send[2], const size_t out[2], void* const recv[2], const size_t in[2]) {
        const int peers[2] = {sc.peer_l, sc.peer_r};
        bool act[2];
        for (int s = 0; s < 2; ++s) {
            act[s] = peers[s] >= 0 && (out[s] > 0 || in[s] > 0);
            REQUIRE(out[s] <= ipc.slot_bytes && in[s] <= ipc.slot_bytes, "IPC link: a message exceeds the mailbox slot");
            if (act[s] && out[s]) HIP_TRY(hipMemcpyAsync(ipc.mailbox + (size_t)s * ipc.slot_bytes, send[s], out[s], hipMemcpyDeviceToDevice, stream));
        }
        HIP_TRY(hipStreamSynchronize(stream));
        for (int s = 0; s < 2; ++s)
            if (act[s] && !sc.self_loop && !ipc.sync_pair(s)) { err = ipc.err; return SMAC_ERR_INVALID; }
        for (int s = 0; s < 2; ++s)        // from the left neighbour its RIGHT slot, from the right neighbour its LEFT slot
            if (act[s] && in[s]) HIP_TRY(hipMemcpyAsync(recv[s], ipc.peer_box[s] + (size_t)(1 - s) * ipc.slot_bytes, in[s], hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (int s = 0; s < 2; ++s)
            if (act[s] && !sc.self_loop && !ipc.sync_pair(s)) { err = ipc.err; return SMAC_ERR_INVALID; }
        return SMAC_OK;
    }
    // 0 (default): the RCCL group is enqueued on the kernels' own stream - in order, no event hand-off.  1: on the communication stream behind
    // two events.  Measured on one GPU (tools/exchange_overhead.py, profiles/r03_h_exchange_overhead.txt): a cross-stream hand-off costs more than
    // the 2 x 0.5 MB exchange itself, and until interior chunks are launched beside it there is nothing for the second stream to overlap with.
    int comm_own_stream = getenv("SMAC_COMM_STREAM") ? atoi(getenv("SMAC_COMM_STREAM")) : 0;
    long long exchanges_done = 0;
    // smac_comm_abort may come from ANOTHER host thread (parallel.FailureWatch: a neighbour failed while this rank's thread sits in a stream
    // synchronisation behind a receive nobody will answer).  The communicator is therefore only touched under this mutex: the enqueueing calls
    // (short, they never wait for the device) and the abort exclude each other; a stream synchronisation is never made while holding it.
    std::mutex comm_mu;
#define NCCL_TRY(expr)                                                                                      \
    do {                                                                                                    \
        ncclResult_t _r = (expr);                                                                           \
        if (_r != ncclSuccess) {                                                                            \
            this->err = std::string(#expr) + " failed: " + Rccl::get().GetErrorString(_r);                 \
            return SMAC_ERR_HIP;                                                                            \
        }                                                                                                   \
    } while (0)
    int comm_init(const char* id128, int rank, int world) override {
        REQUIRE(id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad rank / world / id");
        REQUIRE(!comm, "comm_init: this handle already has a communicator");
        if (comm_stub == 2) {                                 // the IPC link: the id names a shared-memory segment (smac_comm_unique_id made it)
            REQUIRE(!ipc.shm, "comm_init: this handle already has an IPC link");
            HIP_TRY(hipSetDevice(cfg.device));
            if (!ipc.attach(id128, rank, world)) { err = ipc.err; return SMAC_ERR_INVALID; }
            if (!ipc.sync_all()) { err = ipc.err; return SMAC_ERR_INVALID; }
            ipc.unlink_name();                                // every rank holds its mapping: the name goes now, the segment with the last unmap (crashes included)
            c_rank = rank; c_world = world;
            return SMAC_OK;
        }
        Rccl& L = Rccl::get();
        if (!L.load()) { err = L.err; return SMAC_ERR_INVALID; }
        ncclUniqueId id;
        static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes (rccl.h NCCL_UNIQUE_ID_BYTES)");
        memcpy(&id, id128, sizeof id);
        HIP_TRY(hipSetDevice(cfg.device));
        NCCL_TRY(L.CommInitRank(&comm, world, id, rank));
        HIP_TRY(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ev_kernels, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_comm, hipEventDisableTiming));
        c_rank = rank; c_world = world;
        return SMAC_OK;
    }
    // Geometry of this rank's slab: the first shared plane with the left / right neighbour in THIS rank's grid indexing, `nplanes` planes each;
    // contact_*: can a contact primitive reach those planes (parallel.contact_sides, agreed with the neighbour); base_lo..base_hi: stencil bases
    // this rank may hold (own range widened by the drift tolerance).  self_loop (world 1 only): left = right = this rank - what goes out on the
    // left comes in on the right and vice versa (periodic planes): the RCCL path exercised end to end on ONE GPU.
    int comm_slab(int left0, int right0, int nplanes, int contact_left, int contact_right, int base_lo, int base_hi, int self_loop) override {
        REQUIRE(nplanes >= 2 && left0 >= 0 && right0 >= 0 && left0 + nplanes <= D.n && right0 + nplanes <= D.n, "comm_slab: bad plane range");
        REQUIRE(comm || comm_stub, "comm_slab: no communicator (smac_comm_init)");
        REQUIRE(!self_loop || c_world == 1, "comm_slab: self_loop is the world-1 test mode");
        sc.on = true; sc.left0 = left0; sc.right0 = right0; sc.np = nplanes; sc.self_loop = self_loop != 0;
        sc.peer_l = self_loop ? c_rank : (c_rank > 0 ? c_rank - 1 : -1);
        sc.peer_r = self_loop ? c_rank : (c_rank < c_world - 1 ? c_rank + 1 : -1);
        sc.contact_l = contact_left != 0 && sc.peer_l >= 0;
        sc.contact_r = contact_right != 0 && sc.peer_r >= 0;
        D.slab_base_lo = base_lo; D.slab_base_hi = base_hi;
        const size_t rec = (size_t)nplanes * D.n * D.n;
        if (rec != halo_records) {
            HIP_TRY(hipStreamSynchronize(stream));
            hipFree(halo_buf);
            halo_buf = nullptr;
            HIP_TRY(hipMalloc((void**)&halo_buf, 4 * rec * sizeof(Vec4<R>)));
            HIP_TRY(hipMemsetAsync(halo_buf, 0, 4 * rec * sizeof(Vec4<R>), stream));
            halo_records = rec;
        }
        if (comm_stub == 2) {
            REQUIRE(ipc.shm, "comm_slab: no IPC link (smac_comm_init with SMAC_COMM_STUB=2)");
            size_t bytes = rec * sizeof(Vec4<R>);
            if (mig_bytes(cfg.n_particles) > bytes) bytes = mig_bytes(cfg.n_particles);      // (a migration can hand over at most the handle's capacity)
            HIP_TRY(hipStreamSynchronize(stream));
            if (!ipc.open_boxes(bytes, sc.peer_l, sc.peer_r, sc.self_loop)) { err = ipc.err; return SMAC_ERR_HIP; }
        }
        if (!comm_stream) {                                   // (stub mode without a communicator)
            HIP_TRY(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ev_kernels, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ev_comm, hipEventDisableTiming));
        }
        return SMAC_OK;
    }
    // SUM the partials of `field` (minus `minus`) on the shared planes with both neighbours
    // the sides an exchange of this kind has (every boundary, or only those a contact primitive can reach)
    HaloSides halo_sides(bool contact_only, int* peers = nullptr) const {
        HaloSides hs;
        hs.count = 0;
        hs.slot[0] = hs.slot[1] = 0; hs.plane0[0] = hs.plane0[1] = 0;
        if (sc.peer_l >= 0 && (sc.contact_l || !contact_only)) { hs.slot[hs.count] = 0; hs.plane0[hs.count] = sc.left0; if (peers) peers[hs.count] = sc.peer_l; ++hs.count; }
        if (sc.peer_r >= 0 && (sc.contact_r || !contact_only)) { hs.slot[hs.count] = 1; hs.plane0[hs.count] = sc.right0; if (peers) peers[hs.count] = sc.peer_r; ++hs.count; }
        return hs;
    }
    // in_kernels: the producing kernel has packed the planes and the consuming kernel will add what arrives (k_grid_op's two pieces: DevSim::halo_hs)
    int exchange(Vec4<R>* field, const Vec4<R>* minus, bool contact_only, bool in_kernels = false) {
        int peers[2] = {-1, -1};
        const HaloSides hs = halo_sides(contact_only, peers);
        if (hs.count == 0) return SMAC_OK;
        REQUIRE(grid_epoch > 0, "exchange: no epoch bound");
        const size_t rec = halo_records;
        Vec4<R>* send = halo_buf;
        Vec4<R>* recv = halo_buf + 2 * rec;
        if (!in_kernels)
            hipLaunchKernelGGL(k_halo_pack2<R>, dim3(nblk(rec), hs.count), dim3(BLOCK), 0, stream, D, (const Vec4<R>*)field, minus, hs, sc.np, send);   // every side BEFORE any unpack: partials, not totals
        hipStream_t cs = comm_own_stream ? comm_stream : stream;
        if (comm_own_stream) {
            HIP_TRY(hipEventRecord(ev_kernels, stream));
            HIP_TRY(hipStreamWaitEvent(comm_stream, ev_kernels, 0));
        }
        if (comm_stub == 2) {
            const void* snd[2] = {nullptr, nullptr};
            void* rcv[2] = {nullptr, nullptr};
            size_t nb[2] = {0, 0};
            for (int s = 0; s < hs.count; ++s) {
                const int side = hs.slot[s];                  // slot 0 = left neighbour, 1 = right neighbour
                snd[side] = send + (size_t)side * rec; rcv[side] = recv + (size_t)side * rec; nb[side] = rec * sizeof(Vec4<R>);
            }
            int rc = ipc_sendrecv(snd, nb, rcv, nb);
            if (rc) return rc;
        } else if (!comm_stub) {
            Rccl& L = Rccl::get();
            const size_t count = rec * 4;
            std::lock_guard<std::mutex> lock(comm_mu);
            REQUIRE(comm, "exchange: the communicator was aborted (smac_comm_abort)");
            NCCL_TRY(L.GroupStart());
            for (int s = 0; s < hs.count; ++s) NCCL_TRY(L.Send(send + (size_t)hs.slot[s] * rec, count, nccl_type<R>::v, peers[s], comm, cs));
            // self loop: sends and receives between one pair match in order - the first message (sent "to the left") is what a left-hand
            // neighbour's right side would have sent: it arrives in the RIGHT slot, the second in the left one
            for (int s = 0; s < hs.count; ++s) {
                const int into = sc.self_loop ? hs.slot[hs.count - 1 - s] : hs.slot[s];
                NCCL_TRY(L.Recv(recv + (size_t)into * rec, count, nccl_type<R>::v, peers[s], comm, cs));
            }
            NCCL_TRY(L.GroupEnd());
        } else {
            // stub: each side receives what the other side of THIS rank sent (the self-loop's data movement as a device copy)
            for (int s = 0; s < hs.count; ++s)
                HIP_TRY(hipMemcpyAsync(recv + (size_t)hs.slot[hs.count - 1 - s] * rec, send + (size_t)hs.slot[s] * rec, rec * sizeof(Vec4<R>), hipMemcpyDeviceToDevice, cs));
        }
        if (comm_own_stream) {
            HIP_TRY(hipEventRecord(ev_comm, comm_stream));
            HIP_TRY(hipStreamWaitEvent(stream, ev_comm, 0));
        }
        if (!in_kernels)
            hipLaunchKernelGGL(k_halo_unpack_add2<R>, dim3(nblk(rec), hs.count), dim3(BLOCK), 0, stream, D, field, hs, sc.np, (const Vec4<R>*)recv);
        ++exchanges_done;
        return check_launch();
    }
    // A failure on ONE rank inside a collective loop (drift, the slab-range guard, a HIP error ...) returns on that rank only, while its neighbours
    // have the matching ncclRecv enqueued and would wait for ever.  With a live multi-rank communicator the failing rank therefore
    // ABORTS it (ncclCommAbort: its own pending operations are torn down, nothing is retried in-process) and says so in the message; the caller is
    // expected to end the process with a non-zero status so that the launcher tears the peers down (bench.py; parallel.agreed_failure for hosts that
    // have a control plane of their own).  World 1 (the self-loop test mode) has no peer to release and keeps its communicator.
    int slab_guard(int rc) {
        if (rc != SMAC_OK && c_world > 1 && (comm || (comm_stub == 2 && ipc.shm))) {
            {
                std::lock_guard<std::mutex> lock(comm_mu);
                if (comm) {
                    Rccl& L = Rccl::get();
                    if (L.CommAbort) L.CommAbort(comm);
                    comm = nullptr;
                }
            }
            ipc.abort_link();                                 // (IPC test transport: every rank waiting at a barrier of the link returns at once)
            sc.on = false;
            err += " [rank " + std::to_string(c_rank) + ": the communicator was aborted; end this process, or publish the failure (parallel.FailureWatch), so that the other ranks stop]";
        }
        return rc;
    }
    // The host's reaction to ANOTHER rank's failure: no stream sync (a receive may never complete).  Callable from a second host thread while the
    // handle's own thread waits in a synchronisation: ncclCommAbort makes the communicator's device kernels give up, the wait returns, and the next
    // exchange finds no communicator.  Over the IPC test transport the link's abort flag ends any barrier wait.
    int comm_abort() override {
        {
            std::lock_guard<std::mutex> lock(comm_mu);
            if (comm) {
                Rccl& L = Rccl::get();
                if (L.CommAbort) L.CommAbort(comm);
                comm = nullptr;
            }
        }
        ipc.abort_link();
        sc.on = false;
        return SMAC_OK;
    }
    bool halo_in_bwd = false;            // set by smac_substeps_slab_grad: k_reduce_aout packs / k_grid_op_grad adds the shared planes of grid_v_out.grad themselves
    DevSim<R> with_halo(const DevSim<R>& src) const {
        DevSim<R> Dh = src;
        Dh.halo_hs = halo_sides(false);
        Dh.halo_np = sc.np;
        Dh.halo_send = halo_buf;
        Dh.halo_recv = halo_buf + 2 * halo_records;
        return Dh;
    }
    int substeps_slab(int f0, int count) override {
        REQUIRE(sc.on, "substeps_slab: no slab geometry (smac_comm_slab)");
        const bool contact = sc.contact_l || sc.contact_r;
        int rc;
        struct Reset { Sim* s; ~Reset() { s->fwd_hint = -1; s->halo_in_grid_op = false; } } reset_on_exit{this};   // every return: a later single substep must not take the fused path on a stale hint
        for (int f = f0; f < f0 + count; ++f) {
            fwd_hint = f + 1 < f0 + count ? f + 1 : -1;       // lets substep f's G2P piece carry the next substep's P2G (k_g2p_p2g)
            halo_in_grid_op = halo_fuse_env != 0;             // k_grid_op's two pieces pack / add the shared planes of {m,p} themselves (SMAC_HALO_FUSE=0: own launches)
            if ((rc = substep_phase(f, nullptr, 0))) { halo_in_grid_op = false; return slab_guard(rc); }
            if ((rc = exchange(D.vin, nullptr, false, halo_in_grid_op))) { halo_in_grid_op = false; return slab_guard(rc); }   // {m, p} partials after P2G
            rc = substep_phase(f, nullptr, 1);
            halo_in_grid_op = false;
            if (rc) return slab_guard(rc);
            if (contact && any_contact() && (rc = exchange(D.vout, D.vmix, true))) return slab_guard(rc);   // contact corrections v_out - v_mixed
            if ((rc = substep_phase(f, nullptr, 2))) return slab_guard(rc);
        }
        fwd_hint = -1;
        return SMAC_OK;
    }
    int substeps_slab_grad(int f0, int count, const double* ext_f_grad) override {
        REQUIRE(sc.on, "substeps_slab_grad: no slab geometry (smac_comm_slab)");
        const bool contact = sc.contact_l || sc.contact_r;
        int rc;
        struct Reset { Sim* s; ~Reset() { s->bwd_hint = -1; s->halo_in_bwd = false; } } reset_on_exit{this};
        // the exchange of grid_v_out.grad packed by k_reduce_aout and added by k_grid_op_grad (two launches less per substep) - unless a contact primitive can reach a
        // shared plane: the contact adjoint between the two would gather grid_v_out.grad there and needs the totals (SMAC_HALO_FUSE=0: own launches)
        halo_in_bwd = halo_fuse_env != 0 && !(contact && any_contact()) && !(D.collision_type == CONTACT_GRID && any_contact());
        for (int f = f0 + count - 1; f >= f0; --f) {
            bwd_hint = f > f0 ? f - 1 : -1;                   // lets substep f's last piece carry the G2P adjoint of substep f - 1 (k_p2g_g2p_grad)
            if ((rc = substep_grad_phase(f, nullptr, f == f0 + count - 1 ? ext_f_grad : nullptr, nullptr, 0))) { bwd_hint = -1; return slab_guard(rc); }
            if ((rc = exchange(D.aout, nullptr, false, halo_in_bwd))) return slab_guard(rc);          // grid_v_out.grad partials after g2p.grad
            if ((rc = substep_grad_phase(f, nullptr, nullptr, nullptr, 1))) return slab_guard(rc);
            if (contact && any_contact() && (rc = exchange(D.amix, nullptr, true))) return slab_guard(rc);   // grid_v_mixed.grad partials after the contact adjoint
            if ((rc = substep_grad_phase(f, nullptr, nullptr, nullptr, 2))) { bwd_hint = -1; return slab_guard(rc); }
        }
        bwd_hint = -1;
        return SMAC_OK;
    }
    // SUM over the ranks of the per-rank partial wrench sums / primitive-state adjoints, in place - once per env step, where the reference
    // consumes them (rigid_simulator.py:92-93, 203-208): 6 P and 13 P substeps scalars, not grid traffic
    int comm_allreduce(double* buf, size_t n) {
        if (comm_stub == 2 && c_world > 1 && n > 0) {         // IPC link: through the host array of the shared segment, REDUCE_CAP doubles per round
            std::vector<double> mine(IpcLink::REDUCE_CAP);
            for (size_t at = 0; at < n; at += IpcLink::REDUCE_CAP) {
                const size_t m = n - at < IpcLink::REDUCE_CAP ? n - at : IpcLink::REDUCE_CAP;
                HIP_TRY(hipMemcpyAsync(ipc.shm->reduce[c_rank], buf + at, m * sizeof(double), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                if (!ipc.sync_all()) { err = ipc.err; return SMAC_ERR_INVALID; }
                for (size_t i = 0; i < m; ++i) {
                    double acc = 0.0;
                    for (int r = 0; r < c_world; ++r) acc += ipc.shm->reduce[r][i];
                    mine[i] = acc;
                }
                if (!ipc.sync_all()) { err = ipc.err; return SMAC_ERR_INVALID; }          // (everybody has read every slot before the next round overwrites them)
                HIP_TRY(hipMemcpyAsync(buf + at, mine.data(), m * sizeof(double), hipMemcpyHostToDevice, stream));
                HIP_TRY(hipStreamSynchronize(stream));
            }
            return SMAC_OK;
        }
        if (comm_stub || c_world == 1 || n == 0) return SMAC_OK;
        Rccl& L = Rccl::get();
        std::lock_guard<std::mutex> lock(comm_mu);
        REQUIRE(comm, "no communicator (smac_comm_init; or it was aborted)");
        NCCL_TRY(L.AllReduce(buf, buf, n, ncclFloat64, ncclSum, comm, stream));      // on the kernels' stream: ordered with them, nothing to hand over
        return SMAC_OK;
    }
    int comm_allreduce_ext_f(double* total_out, int clear) override {
        const size_t n = (size_t)D.P * 6;
        int rc = comm_allreduce(D.ext_f, n);
        if (rc) return rc;
        if (total_out && n) HIP_TRY(hipMemcpyAsync(total_out, D.ext_f, n * sizeof(double), hipMemcpyDeviceToHost, stream));
        if (clear && n) HIP_TRY(hipMemsetAsync(D.ext_f, 0, n * sizeof(double), stream));
        // the sheet's per-vertex force (soft_cloth primitive_cloth.py:274-278) is a per-rank partial sum too; it stays on the device (read with
        // smac_cloth_get_ext_f, cleared with smac_cloth_clear_ext_f as in the single-domain loop)
        if (D.cloth.present && (rc = comm_allreduce(D.cloth.ext_f, (size_t)D.cloth.V * 3))) return rc;
        if (total_out) HIP_TRY(hipStreamSynchronize(stream));
        return SMAC_OK;
    }
    int comm_allreduce_prim_grad(int f0, int f1) override {
        REQUIRE(f0 >= 0 && f0 < f1 && f1 <= cfg.max_frames, "comm_allreduce_prim_grad: bad frame range");
        for (int i = 0; i < D.P; ++i) {
            int rc = comm_allreduce(pgrad(i) + (size_t)f0 * 13, (size_t)(f1 - f0) * 13);
            if (rc) return rc;
        }
        if (D.cloth.present && D.cloth.pos_grad) {             // the sheet's vertex adjoints of those frames
            const size_t at = (size_t)f0 * D.cloth.V * 3, n = (size_t)(f1 - f0) * D.cloth.V * 3;
            int rc;
            if ((rc = comm_allreduce(D.cloth.pos_grad + at, n)) || (rc = comm_allreduce(D.cloth.vel_grad + at, n))) return rc;
        }
        return SMAC_OK;
    }
    int comm_destroy() override {
        if (stream) HIP_TRY(hipStreamSynchronize(stream));
        if (comm_stream) HIP_TRY(hipStreamSynchronize(comm_stream));
        if (comm) { Rccl::get().CommDestroy(comm); comm = nullptr; }
        ipc.detach();
        if (comm_stream) { hipStreamDestroy(comm_stream); comm_stream = nullptr; }
        if (ev_kernels) { hipEventDestroy(ev_kernels); ev_kernels = nullptr; }
        if (ev_comm) { hipEventDestroy(ev_comm); ev_comm = nullptr; }
        hipFree(halo_buf); halo_buf = nullptr; halo_records = 0;
        sc = SlabCfg();
        D.slab_base_lo = 1; D.slab_base_hi = 0;
        return SMAC_OK;
    }

    // ---- particle migration between slabs, on the device (smac_migrate.hpp) ------------------------------------------------------
    struct MigRec { int frame, epoch, n_old, nk, nl_out, nr_out, nl_in, nr_in; int* src_slot; long long* ids_old; };
    std::vector<MigRec> migs;
    long long* d_ids = nullptr;         // global particle ids of the current segment, in its caller (identity) order
    int *mig_flags = nullptr, *mig_pos = nullptr, *mig_counts = nullptr;     // 3 x Npad flags, 3 x Npad positions, counts exchanged with the neighbours
    char* mig_buf[4] = {nullptr, nullptr, nullptr, nullptr};                 // send L, send R, recv L, recv R
    int mig_cap[4] = {0, 0, 0, 0};
    size_t mig_bytes(int cap) const { return (size_t)cap * (sizeof(long long) + NCOMP * sizeof(R)); }
    int mig_reserve(int which, int cap) {
        if (cap <= mig_cap[which]) return SMAC_OK;
        HIP_TRY(hipStreamSynchronize(stream));
        hipFree(mig_buf[which]);
        mig_buf[which] = nullptr;
        cap = cap + cap / 4 + 1024;
        HIP_TRY(hipMalloc((void**)&mig_buf[which], mig_bytes(cap)));
        mig_cap[which] = cap;
        return SMAC_OK;
    }
    int ensure_ids() {
        if (d_ids) return SMAC_OK;
        HIP_TRY(hipMalloc((void**)&d_ids, (size_t)D.Npad * sizeof(long long)));
        std::vector<long long> h(D.Npad);
        for (int i = 0; i < D.Npad; ++i) h[i] = i;
        HIP_TRY(hipMemcpy(d_ids, h.data(), h.size() * sizeof(long long), hipMemcpyHostToDevice));
        return SMAC_OK;
    }
    int set_ids(const int64_t* ids) override {
        REQUIRE(ids, "null argument");
        int rc = ensure_ids();
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(d_ids, ids, (size_t)D.N * sizeof(long long), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SMAC_OK;
    }
    int get_ids(int64_t* ids) override {
        REQUIRE(ids, "null argument");
        int rc = ensure_ids();
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(ids, d_ids, (size_t)D.N * sizeof(long long), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SMAC_OK;
    }
    // pairwise exchange with the two neighbours in one RCCL group: `bytes_out[s]` from send[s], `bytes_in[s]` into recv[s] (s = 0 left, 1 right);
    // self loop: what leaves on the left arrives on the right and vice versa (the stub moves it with device copies)
    int neighbour_exchange(const void* const send[2], const size_t bytes_out[2], void* const recv[2], const size_t bytes_in[2]) {
        const int peers[2] = {sc.peer_l, sc.peer_r};
        if (comm_stub == 2) return ipc_sendrecv(send, bytes_out, recv, bytes_in);
        if (comm_stub) {
            for (int s = 0; s < 2; ++s)
                if (peers[s] >= 0 && bytes_out[s]) HIP_TRY(hipMemcpyAsync(recv[1 - s], send[s], bytes_out[s], hipMemcpyDeviceToDevice, stream));
            return SMAC_OK;
        }
        Rccl& L = Rccl::get();
        std::lock_guard<std::mutex> lock(comm_mu);
        REQUIRE(comm, "no communicator (smac_comm_init; or it was aborted)");
        NCCL_TRY(L.GroupStart());
        for (int s = 0; s < 2; ++s)
            if (peers[s] >= 0 && bytes_out[s]) NCCL_TRY(L.Send(send[s], bytes_out[s], ncclUint8, peers[s], comm, stream));
        for (int s = 0; s < 2; ++s) {
            const int into = sc.self_loop ? 1 - s : s;
            if (peers[s] >= 0 && bytes_in[into]) NCCL_TRY(L.Recv(recv[into], bytes_in[into], ncclUint8, peers[s], comm, stream));
        }
        NCCL_TRY(L.GroupEnd());
        return SMAC_OK;
    }
    // Hand the particles of frame f whose stencil base left [base_lo, base_hi) to the neighbour on that side; frame f + 1 starts the next segment
    // (kept particles, then the arrivals from the left, then from the right).  out3: {live particles now, sent away, received}.
    int migrate(int f, int base_lo, int base_hi, int32_t* out3) override {
        MigRec rec = {};
        bool filed = false;
        const int rc = migrate_impl(f, base_lo, base_hi, out3, rec, filed);
        if (rc && !filed) { hipFree(rec.src_slot); hipFree(rec.ids_old); }       // (the record's buffers leaked on the error paths behind their allocation)
        return slab_guard(rc);
    }
    int migrate_impl(int f, int base_lo, int base_hi, int32_t* out3, MigRec& rec, bool& filed) {
        int rc;
        REQUIRE(sc.on, "migrate: no slab geometry (smac_comm_slab)");
        REQUIRE(f >= 0 && f + 1 < cfg.max_frames && frame_epoch[f] >= 0, "migrate: frame f holds no state or f + 1 exceeds max_frames");
        REQUIRE(!rolling(), "migrate: not with rolling adjoint storage");
        if ((rc = check_drift()) || (rc = ensure_ids())) return rc;
        if (!mig_flags) {
            HIP_TRY(hipMalloc((void**)&mig_flags, 3 * (size_t)D.Npad * sizeof(int)));
            HIP_TRY(hipMalloc((void**)&mig_pos, 3 * (size_t)D.Npad * sizeof(int)));
            HIP_TRY(hipMalloc((void**)&mig_counts, 8 * sizeof(int)));
        }
        const int e = frame_epoch[f], N = D.N, Npad = D.Npad;
        const R* Sf = D.S + (size_t)f * frame_scalars();
        R* Sn = D.S + (size_t)(f + 1) * frame_scalars();
        int *keep = mig_flags, *left = mig_flags + Npad, *right = mig_flags + 2 * Npad;
        int *kpos = mig_pos, *lpos = mig_pos + Npad, *rpos = mig_pos + 2 * Npad;
        hipLaunchKernelGGL(k_mig_classify<R>, dim3(nblk(N)), dim3(BLOCK), 0, stream, N, Npad, D.n, Sf, base_lo, base_hi, sc.peer_l >= 0 ? 1 : 0, sc.peer_r >= 0 ? 1 : 0,
                           keep, left, right);
        if ((rc = scan(keep, kpos, N)) || (rc = scan(left, lpos, N)) || (rc = scan(right, rpos, N))) return rc;
        int last[6];                                                        // last flag + last position of each class -> the three totals
        for (int k = 0; k < 3; ++k) {
            HIP_TRY(hipMemcpyAsync(&last[2 * k], mig_flags + (size_t)k * Npad + (N - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(&last[2 * k + 1], mig_pos + (size_t)k * Npad + (N - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
        }
        HIP_TRY(hipStreamSynchronize(stream));
        const int nk = last[0] + last[1], nl = last[2] + last[3], nr = last[4] + last[5];
        // counts to / from the neighbours
        int out_counts[2] = {nl, nr}, in_counts[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(mig_counts, out_counts, 2 * sizeof(int), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(mig_counts + 2, 0, 2 * sizeof(int), stream));
        {
            const void* snd[2] = {mig_counts, mig_counts + 1};
            void* rcv[2] = {mig_counts + 2, mig_counts + 3};
            const size_t four[2] = {sizeof(int), sizeof(int)};
            if ((rc = neighbour_exchange(snd, four, rcv, four))) return rc;
        }
        HIP_TRY(hipMemcpyAsync(in_counts, mig_counts + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        const int n_new = nk + in_counts[0] + in_counts[1];
        REQUIRE(n_new >= 1, "migrate: this slab lost all its particles");
        REQUIRE(n_new <= cfg.n_particles, "migrate: more particles arrive than the handle's capacity (n_particles) holds");
        if ((rc = mig_reserve(0, nl)) || (rc = mig_reserve(1, nr)) || (rc = mig_reserve(2, in_counts[0])) || (rc = mig_reserve(3, in_counts[1]))) return rc;
        rec = MigRec{f, e, N, nk, nl, nr, in_counts[0], in_counts[1], nullptr, nullptr};
        HIP_TRY(hipMalloc((void**)&rec.src_slot, (size_t)(N > 0 ? N : 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void**)&rec.ids_old, (size_t)Npad * sizeof(long long)));
        HIP_TRY(hipMemcpyAsync(rec.ids_old, d_ids, (size_t)Npad * sizeof(long long), hipMemcpyDeviceToDevice, stream));
        long long* ids_new = (long long*)tmp_frame;                      // (scratch: Npad x 8 bytes fit a frame of 24 x Npad scalars)
        hipLaunchKernelGGL(k_mig_pack<R>, dim3(nblk(N)), dim3(BLOCK), 0, stream, N, Npad, Sf, Sn, (const long long*)rec.ids_old, ids_new,
                           e > 0 ? (const int*)epochs[e].orig : (const int*)nullptr, (const int*)keep, (const int*)kpos, (const int*)lpos, (const int*)rpos,
                           (const int*)left, mig_buf[0], mig_buf[1], nl > 0 ? nl : 1, nr > 0 ? nr : 1, nk, nl, rec.src_slot);   // (buffer layout [ids: count x 8][rows: 24 x count])
        {
            const void* snd[2] = {mig_buf[0], mig_buf[1]};
            void* rcv[2] = {mig_buf[2], mig_buf[3]};
            const size_t bo[2] = {mig_bytes(nl), mig_bytes(nr)}, bi[2] = {mig_bytes(in_counts[0]), mig_bytes(in_counts[1])};
            if ((rc = neighbour_exchange(snd, bo, rcv, bi))) return rc;
        }
        if (in_counts[0]) hipLaunchKernelGGL(k_mig_unpack<R>, dim3(nblk(in_counts[0])), dim3(BLOCK), 0, stream, in_counts[0], in_counts[0], Npad, (const char*)mig_buf[2], Sn, ids_new, nk);
        if (in_counts[1]) hipLaunchKernelGGL(k_mig_unpack<R>, dim3(nblk(in_counts[1])), dim3(BLOCK), 0, stream, in_counts[1], in_counts[1], Npad, (const char*)mig_buf[3], Sn, ids_new,
                                             nk + in_counts[0]);
        hipLaunchKernelGGL(k_mig_pad<R>, dim3(nblk(Npad - n_new + 1)), dim3(BLOCK), 0, stream, n_new, Npad, Sn);
        HIP_TRY(hipMemcpyAsync(d_ids, ids_new, (size_t)n_new * sizeof(long long), hipMemcpyDeviceToDevice, stream));
        migs.push_back(rec);
        filed = true;
        D.N = n_new;
        D.frame_shift += 1;
        frame_epoch[f + 1] = 0;                                            // the new segment's caller order; binned at its first substep
        ck_epoch[f + 1] = -1;
        fwd_head = f + 1;
        if (out3) { out3[0] = n_new; out3[1] = nl + nr; out3[2] = in_counts[0] + in_counts[1]; }
        return check_launch();
    }
    // Backward of the most recent migrate: the adjoint of the later segment's first frame goes back to the frame it was copied from - across the
    // slab boundary for the particles that crossed it - and is ADDED there (frame f may carry seeds of its own).
    int migrate_grad() override {
        // the segment bookkeeping (live particle count, frame shift) is restored when the way back fails: the record is still on the tape then,
        // and the handle must keep describing the LATER segment it belongs to
        const int n_keep = D.N, shift_keep = D.frame_shift;
        const int rc = migrate_grad_impl();
        if (rc) { D.N = n_keep; D.frame_shift = shift_keep; }
        return slab_guard(rc);
    }
    int migrate_grad_impl() {
        int rc;
        REQUIRE(!migs.empty(), "migrate_grad: no migration on record");
        if ((rc = need_grad())) return rc;
        MigRec rec = migs.back();
        const int f = rec.frame, Npad = D.Npad;
        // adjoint of frame f + 1 in the segment's identity order
        const R* G = nullptr;
        if (adj_epoch[f + 1] < 0 && (rc = adj_make_zero(f + 1))) return rc;
        if ((rc = adjoint_in_order(f + 1, 0, &G))) return rc;
        // frame f's adjoint in the order of frame f's state (epoch rec.epoch); frame f belongs to the EARLIER segment
        D.N = rec.n_old;
        D.frame_shift -= 1;
        if (adj_epoch[f] < 0) {
            if ((rc = adj_make_zero(f))) return rc;
        } else if (adj_epoch[f] != rec.epoch) {
            const R* tmp = nullptr;
            if (!tmp_frame2) HIP_TRY(hipMalloc((void**)&tmp_frame2, frame_scalars() * sizeof(R)));
            if ((rc = adjoint_in_order(f, rec.epoch, &tmp, G == tmp_frame ? tmp_frame2 : tmp_frame))) return rc;
            HIP_TRY(hipMemcpyAsync(adj_ptr(f), tmp, frame_scalars() * sizeof(R), hipMemcpyDeviceToDevice, stream));
        }
        adj_epoch[f] = rec.epoch;
        adj_stale[f] = 0;
        R* Af = adj_ptr(f);
        if (rec.nk) hipLaunchKernelGGL(k_mig_grad_keep<R>, dim3(nblk(rec.nk)), dim3(BLOCK), 0, stream, rec.nk, Npad, G, Af, (const int*)rec.src_slot);
        // the arrivals' adjoint rows go back to their senders; the rows of the particles this rank sent away come back
        if ((rc = mig_reserve(0, rec.nl_in)) || (rc = mig_reserve(1, rec.nr_in)) || (rc = mig_reserve(2, rec.nl_out)) || (rc = mig_reserve(3, rec.nr_out))) return rc;
        if (rec.nl_in) hipLaunchKernelGGL(k_mig_grad_pack<R>, dim3(nblk(rec.nl_in)), dim3(BLOCK), 0, stream, rec.nl_in, rec.nl_in, Npad, G, rec.nk, mig_buf[0]);
        if (rec.nr_in) hipLaunchKernelGGL(k_mig_grad_pack<R>, dim3(nblk(rec.nr_in)), dim3(BLOCK), 0, stream, rec.nr_in, rec.nr_in, Npad, G, rec.nk + rec.nl_in, mig_buf[1]);
        {
            const void* snd[2] = {mig_buf[0], mig_buf[1]};
            void* rcv[2] = {mig_buf[2], mig_buf[3]};
            const size_t bo[2] = {mig_bytes(rec.nl_in), mig_bytes(rec.nr_in)}, bi[2] = {mig_bytes(rec.nl_out), mig_bytes(rec.nr_out)};
            if ((rc = neighbour_exchange(snd, bo, rcv, bi))) return rc;
        }
        if (rec.nl_out) hipLaunchKernelGGL(k_mig_grad_unpack<R>, dim3(nblk(rec.nl_out)), dim3(BLOCK), 0, stream, rec.nl_out, rec.nl_out, Npad, (const char*)mig_buf[2], Af,
                                           (const int*)(rec.src_slot + rec.nk));
        if (rec.nr_out) hipLaunchKernelGGL(k_mig_grad_unpack<R>, dim3(nblk(rec.nr_out)), dim3(BLOCK), 0, stream, rec.nr_out, rec.nr_out, Npad, (const char*)mig_buf[3], Af,
                                           (const int*)(rec.src_slot + rec.nk + rec.nl_out));
        HIP_TRY(hipMemcpyAsync(d_ids, rec.ids_old, (size_t)Npad * sizeof(long long), hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        hipFree(rec.src_slot); hipFree(rec.ids_old);
        migs.pop_back();
        return check_launch();
    }
    bool own_stream = true;
    int set_stream(void* s) override {
        HIP_TRY(hipStreamSynchronize(stream));
        if (own_stream && stream) hipStreamDestroy(stream);
        stream = (hipStream_t)s;
        own_stream = false;
        return SMAC_OK;
    }

    // ---- primitives -----------------------------------------------------------------------
    int check_prim(int prim) {
        REQUIRE(prim >= 0 && prim < D.P, "primitive index out of range");
        return SMAC_OK;
    }
    double* pstate(int prim) { return D.prim_state + (size_t)prim * cfg.max_frames * 13; }
    double* pgrad(int prim) { return D.prim_grad + (size_t)prim * cfg.max_frames * 13; }
    int prim_upload_sdf(int prim, const double* sdf, const double* normal, const int32_t* res, const double* lower,
                        const double* upper, double dx) override {
        int rc = check_prim(prim);
        if (rc) return rc;
        REQUIRE(sdf && normal && res && lower && upper && dx > 0, "null table");
        REQUIRE(res[0] >= 2 && res[1] >= 2 && res[2] >= 2, "sdf res < 2");
        const size_t cells = (size_t)res[0] * res[1] * res[2];
        HIP_TRY(hipStreamSynchronize(stream));
        if ((void*)prim_tables64[prim][0] != (void*)prim_tables[prim][0]) { hipFree(prim_tables64[prim][0]); hipFree(prim_tables64[prim][1]); }
        hipFree(prim_tables[prim][0]); hipFree(prim_tables[prim][1]);
        prim_tables[prim][0] = prim_tables[prim][1] = nullptr;
        prim_tables64[prim][0] = prim_tables64[prim][1] = nullptr;
        // f64 tables for the forecast contact chain; in f32 mode a float copy serves the band filter and collision types 0 / 1
        HIP_TRY(hipMalloc((void**)&prim_tables64[prim][0], cells * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&prim_tables64[prim][1], cells * 3 * sizeof(double)));
        HIP_TRY(hipMemcpy(prim_tables64[prim][0], sdf, cells * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(prim_tables64[prim][1], normal, cells * 3 * sizeof(double), hipMemcpyHostToDevice));
        if (sizeof(R) == sizeof(double)) {
            prim_tables[prim][0] = (R*)prim_tables64[prim][0];
            prim_tables[prim][1] = (R*)prim_tables64[prim][1];
        } else {
            std::vector<R> ts(cells), tn(cells * 3);
            for (size_t i = 0; i < cells; ++i) ts[i] = (R)sdf[i];
            for (size_t i = 0; i < cells * 3; ++i) tn[i] = (R)normal[i];
            HIP_TRY(hipMalloc((void**)&prim_tables[prim][0], cells * sizeof(R)));
            HIP_TRY(hipMalloc((void**)&prim_tables[prim][1], cells * 3 * sizeof(R)));
            HIP_TRY(hipMemcpy(prim_tables[prim][0], ts.data(), cells * sizeof(R), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(prim_tables[prim][1], tn.data(), cells * 3 * sizeof(R), hipMemcpyHostToDevice));
        }
        PrimTable<R>& T = D.prim[prim];
        PrimTable<double>& T64 = D.prim64[prim];
        T.sdf = prim_tables[prim][0]; T.normal = prim_tables[prim][1];
        T64.sdf = prim_tables64[prim][0]; T64.normal = prim_tables64[prim][1];
        for (int d = 0; d < 3; ++d) {
            T.res[d] = res[d]; T.lower[d] = (R)lower[d]; T.upper[d] = (R)upper[d];
            T64.res[d] = res[d]; T64.lower[d] = lower[d]; T64.upper[d] = upper[d];
        }
        T.inv_dx = (R)(1.0 / dx);                                             // mesh.py:29
        T64.inv_dx = 1.0 / dx;
        ++config_gen;
        return SMAC_OK;
    }
    int prim_set_params(int prim, double friction, double softness, int contact) override {
        int rc = check_prim(prim);
        if (rc) return rc;
        D.prim[prim].friction = (R)friction; D.prim[prim].softness = (R)softness; D.prim[prim].contact = contact ? 1 : 0;
        D.prim64[prim].friction = friction; D.prim64[prim].softness = softness; D.prim64[prim].contact = contact ? 1 : 0;
        ++config_gen;
        return SMAC_OK;
    }
    int prim_set_state(int prim, int f0, int f1, const double* s13) override {
        int rc = check_prim(prim);
        if (rc) return rc;
        REQUIRE(s13 && f0 >= 0 && f1 <= cfg.max_frames && f0 < f1, "prim_set_state: bad frame range");
        for (int f = f0; f < f1; ++f) ck_epoch[f] = -1;
        std::vector<double> tmp((size_t)(f1 - f0) * 13);
        for (int f = 0; f < f1 - f0; ++f)
            for (int c = 0; c < 13; ++c) tmp[(size_t)f * 13 + c] = s13[c];
        HIP_TRY(hipMemcpyAsync(pstate(prim) + (size_t)f0 * 13, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SMAC_OK;
    }
    int prim_get_state(int prim, int f, double* s13) override {
        int rc;
        if ((rc = check_prim(prim)) || (rc = check_frame(f))) return rc;
        HIP_TRY(hipMemcpyAsync(s13, pstate(prim) + (size_t)f * 13, 13 * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SMAC_OK;
    }
    int prim_get_state_grad(int prim, int f0, int f1, double* g13) override {
        int rc = check_prim(prim);
        if (rc) return rc;
        REQUIRE(g13 && f0 >= 0 && f1 <= cfg.max_frames && f0 < f1, "prim_get_state_grad: bad frame range");
        std::vector<double> tmp((size_t)(f1 - f0) * 13);
        HIP_TRY(hipMemcpyAsync(tmp.data(), pgrad(prim) + (size_t)f0 * 13, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (int c = 0; c < 13; ++c) g13[c] = 0;
        for (int f = 0; f < f1 - f0; ++f)
            for (int c = 0; c < 13; ++c) g13[c] += tmp[(size_t)f * 13 + c];
        return SMAC_OK;
    }
    int prim_set_states(int prim, int f0, int f1, const double* s13) override {       // one state per frame of [f0, f1)
        int rc = check_prim(prim);
        if (rc) return rc;
        REQUIRE(s13 && f0 >= 0 && f1 <= cfg.max_frames && f0 < f1, "prim_set_states: bad frame range");
        for (int f = f0; f < f1; ++f) ck_epoch[f] = -1;
        HIP_TRY(hipMemcpyAsync(pstate(prim) + (size_t)f0 * 13, s13, (size_t)(f1 - f0) * 13 * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));                                          // (the caller's array may go away)
        return SMAC_OK;
    }
    int prim_get_state_grads(int prim, int f0, int f1, double* g13) override {
        int rc = check_prim(prim);
        if (rc) return rc;
        REQUIRE(g13 && f0 >= 0 && f1 <= cfg.max_frames && f0 < f1, "prim_get_state_grads: bad frame range");
        HIP_TRY(hipMemcpyAsync(g13, pgrad(prim) + (size_t)f0 * 13, (size_t)(f1 - f0) * 13 * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SMAC_OK;
    }
    int prim_add_state_grad(int prim, int f, const double* g13) override {
        int rc;
        if ((rc = check_prim(prim)) || (rc = check_frame(f))) return rc;
        double tmp[13];
        HIP_TRY(hipMemcpyAsync(tmp, pgrad(prim) + (size_t)f * 13, sizeof tmp, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (int c = 0; c < 13; ++c) tmp[c] += g13[c];
        HIP_TRY(hipMemcpyAsync(pgrad(prim) + (size_t)f * 13, tmp, sizeof tmp, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SMAC_OK;
    }
    int prim_fk(int prim, int f) override {
        int rc = check_prim(prim);
        if (rc) return rc;
        REQUIRE(f >= 0 && f + 1 < cfg.max_frames, "forward_kinematics: frame out of range");
        prof_begin(K_FK);
        hipLaunchKernelGGL(k_prim_fk<double>, dim3(1), dim3(64), 0, stream, pstate(prim), f, D.dt64, 1, (size_t)0);
        prof_end();
        return check_launch();
    }
    int prim_fk_grad(int prim, int f) override {
        int rc = check_prim(prim);
        if (rc) return rc;
        REQUIRE(f >= 0 && f + 1 < cfg.max_frames, "forward_kinematics.grad: frame out of range");
        prof_begin(K_FK);
        hipLaunchKernelGGL(k_prim_fk_grad<double>, dim3(1), dim3(64), 0, stream, (const double*)pstate(prim), pgrad(prim), f, D.dt64, (size_t)0);
        prof_end();
        return check_launch();
    }
    int prim_get_ext_f(int prim, double* e6) override {
        int rc = check_prim(prim);
        if (rc || (rc = check_drift())) return rc;             // a drifted epoch is repaired BEFORE the host consumes its wrench
        HIP_TRY(hipMemcpyAsync(e6, D.ext_f + prim * 6, 6 * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SMAC_OK;
    }
    int prim_clear_ext_f(int prim) override {                                 // :183-187 (value and grad)
        int rc = check_prim(prim);
        if (rc) return rc;
        if (fwd_head >= 0 && !bwd_since_fwd && (rc = check_drift())) return rc;   // (repairs with the accumulator as the forward pass left it)
        HIP_TRY(hipMemsetAsync(D.ext_f + prim * 6, 0, 6 * sizeof(double), stream));
        HIP_TRY(hipMemsetAsync(D.ext_f_grad + prim * 6, 0, 6 * sizeof(double), stream));
        // a later repair of this epoch re-accumulates from here, not from the epoch's first frame (no double counting)
        if (ext_snap && fwd_head >= 0 && !repairing) return snapshot_ext(fwd_head);
        return SMAC_OK;
    }
    // velocity control (primitive_base.py:285-319): action_buffer[s] = a6 ; v[j] = a[3:6], w[j] = a[0:3] for j in [s n, (s+1) n)
    int prim_set_action(int prim, int s, int n, const double* a6) override {
        int rc = check_prim(prim);
        if (rc) return rc;
        REQUIRE(a6 && s >= 0 && n >= 1 && n <= 1024 && (s + 1) * n <= cfg.max_frames, "prim_set_action: frames out of range");
        for (int f = s * n; f < (s + 1) * n; ++f) ck_epoch[f] = -1;
        Act6 a;
        for (int c = 0; c < 6; ++c) a.a[c] = a6[c];
        // on the device, the action in the kernel's argument block: no staging copy, no stream sync per primitive and env step
        hipLaunchKernelGGL(k_prim_set_action, dim3(1), dim3(n < 64 ? 64 : (n + 63) / 64 * 64), 0, stream, pstate(prim),
                           action_buf + (size_t)prim * cfg.max_frames * 6, s, n, a);
        return check_launch();
    }
    int prim_get_action_grad(int prim, int s, int n, double* g6) override {
        int rc = check_prim(prim);
        if (rc) return rc;
        REQUIRE(g6 && s >= 0 && n >= 1 && (s + 1) * n <= cfg.max_frames, "prim_get_action_grad: frames out of range");
        double* abg = action_buf_grad + (size_t)prim * cfg.max_frames * 6;
        hipLaunchKernelGGL(k_prim_action_grad, dim3(1), dim3(64), 0, stream, (const double*)pgrad(prim), abg, s, n);
        HIP_TRY(hipMemcpyAsync(g6, abg + (size_t)s * 6, 6 * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return check_launch();
    }
    int prim_get_action_grads(int prim, int s0, int s1, int n, double* g6) override {   // the env steps [s0, s1) of an episode at once
        int rc = check_prim(prim);
        if (rc) return rc;
        REQUIRE(g6 && s0 >= 0 && s0 < s1 && n >= 1 && (size_t)s1 * n <= (size_t)cfg.max_frames, "prim_get_action_grads: frames out of range");
        double* abg = action_buf_grad + (size_t)prim * cfg.max_frames * 6;
        hipLaunchKernelGGL(k_prim_action_grad, dim3(s1 - s0), dim3(64), 0, stream, (const double*)pgrad(prim), abg, s0, n);
        HIP_TRY(hipMemcpyAsync(g6, abg + (size_t)s0 * 6, (size_t)(s1 - s0) * 6 * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return check_launch();
    }
    int prim_reset(int prim) override {                                       // :271-275
        int rc = check_prim(prim);
        if (rc) return rc;
        ++config_gen;
        HIP_TRY(hipMemsetAsync(pstate(prim), 0, (size_t)cfg.max_frames * 13 * sizeof(double), stream));
        HIP_TRY(hipMemsetAsync(pgrad(prim), 0, (size_t)cfg.max_frames * 13 * sizeof(double), stream));
        HIP_TRY(hipMemsetAsync(action_buf + (size_t)prim * cfg.max_frames * 6, 0, (size_t)cfg.max_frames * 6 * sizeof(double), stream));
        HIP_TRY(hipMemsetAsync(action_buf_grad + (size_t)prim * cfg.max_frames * 6, 0, (size_t)cfg.max_frames * 6 * sizeof(double), stream));
        return prim_clear_ext_f(prim);
    }

    // ---- measurement ----------------------------------------------------------------------
    int timer_start() override { HIP_TRY(hipEventRecord(t0, stream)); return SMAC_OK; }
    int timer_stop(double* ms) override {
        HIP_TRY(hipEventRecord(t1, stream));
        HIP_TRY(hipEventSynchronize(t1));
        float f = 0;
        HIP_TRY(hipEventElapsedTime(&f, t0, t1));
        if (ms) *ms = f;
        return SMAC_OK;
    }
    int profile_enable(int on) override { profiling = on != 0; return SMAC_OK; }
    int profile_reset() override {
        int rc = prof_collect();
        for (int i = 0; i < K_COUNT; ++i) { prof_ms[i] = 0; prof_n[i] = 0; }
        return rc;
    }
    int profile_get(int i, char* name, int cap, double* ms, int64_t* launches) override {
        REQUIRE(i >= 0 && i < K_COUNT, "profile index out of range");
        int rc = prof_collect();
        if (rc) return rc;
        if (name && cap > 0) { strncpy(name, kKernelNames[i], cap - 1); name[cap - 1] = 0; }
        if (ms) *ms = prof_ms[i];
        if (launches) *launches = prof_n[i];
        return SMAC_OK;
    }
    int grid_ptr(const char* field, void** p, int64_t* n, int32_t* bytes) override {
        REQUIRE(field && p, "null argument");
        // 4-scalar records per cell, block-major (smac_sort.hpp): {m,p} / {v_mixed,0} / {v_out,0} and adjoints
        struct { const char* name; R* ptr; int64_t n; } tab[] = {
            {"grid_in", (R*)D.vin, (int64_t)(4 * D.G)}, {"grid_mixed", (R*)D.vmix, (int64_t)(4 * D.G)}, {"grid_out", (R*)D.vout, (int64_t)(4 * D.G)},
            {"grid_in.grad", (R*)D.ain, (int64_t)(4 * D.G)}, {"grid_mixed.grad", (R*)D.amix, (int64_t)(4 * D.G)},
            {"grid_out.grad", (R*)D.aout, (int64_t)(4 * D.G)},
            {"state", D.S, (int64_t)frame_scalars() * cfg.max_frames}, {"state.grad", D.A, (int64_t)frame_scalars() * adj_slots},
            {"slab", (R*)slab, (int64_t)(slab_chunks * TILE_WORDS * 4)}};
        for (auto& t : tab)
            if (!strcmp(t.name, field)) {
                *p = t.ptr;
                if (n) *n = t.n;
                if (bytes) *bytes = (int32_t)sizeof(R);
                return SMAC_OK;
            }
        // primitive accumulators (always f64): ext_f[P][6], state.grad[P][max_frames][13] - what the slabs all-reduce per env step
        const int Pn = D.P > 0 ? D.P : 1;
        struct { const char* name; double* ptr; int64_t n; } tab64[] = {
            {"ext_f", D.ext_f, (int64_t)Pn * 6}, {"prim_state.grad", D.prim_grad, (int64_t)Pn * cfg.max_frames * 13}};
        for (auto& t : tab64)
            if (!strcmp(t.name, field)) {
                *p = t.ptr;
                if (n) *n = t.n;
                if (bytes) *bytes = 8;
                return SMAC_OK;
            }
        err = std::string("unknown grid field ") + field;
        return SMAC_ERR_INVALID;
    }
    int stream_handle(void** s) override { *s = (void*)stream; return SMAC_OK; }
};

struct smac_sim { ISim* impl; };

extern "C" {

const char* smac_last_error(smac_handle h) { return h ? h->impl->err.c_str() : g_create_error.c_str(); }
int smac_abi_version(void) { return SMAC_ABI_VERSION; }
int smac_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int smac_mesh_to_sdf(int device, const double* vertices, int nv, const int32_t* faces, int nf, const double lower[3],
                     const int32_t res[3], double dx, double* sdf_out, double* normal_out) {
    if (!vertices || !faces || !lower || !res || !sdf_out || !normal_out) { g_create_error = "null argument"; return SMAC_ERR_INVALID; }
    if (nv <= 0 || nf <= 0 || res[0] <= 0 || res[1] <= 0 || res[2] <= 0 || !(dx > 0)) { g_create_error = "empty mesh or grid"; return SMAC_ERR_INVALID; }
    const int ndev = smac_device_count();
    if (ndev <= 0) { g_create_error = "no HIP device visible: libsoftmac_hip has no CPU fallback"; return SMAC_ERR_NOGPU; }
    if (device < 0 || device >= ndev) { g_create_error = "device ordinal out of range"; return SMAC_ERR_INVALID; }
    std::vector<double> tri((size_t)nf * 9);
    for (int t = 0; t < nf; ++t)
        for (int c = 0; c < 3; ++c) {
            const int v = faces[3 * t + c];
            if (v < 0 || v >= nv) { g_create_error = "face index out of range"; return SMAC_ERR_INVALID; }
            for (int d = 0; d < 3; ++d) tri[(size_t)t * 9 + 3 * c + d] = vertices[3 * (size_t)v + d];
        }
    const size_t total = (size_t)res[0] * res[1] * res[2];
    double *d_tri = nullptr, *d_sdf = nullptr, *d_nrm = nullptr;
    auto fail = [&](hipError_t e, const char* what) {
        g_create_error = std::string(what) + " failed: " + hipGetErrorString(e);
        hipFree(d_tri); hipFree(d_sdf); hipFree(d_nrm);
        return SMAC_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fail(e, "hipSetDevice");
    if ((e = hipMalloc((void**)&d_tri, tri.size() * sizeof(double))) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&d_sdf, total * sizeof(double))) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&d_nrm, total * 3 * sizeof(double))) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMemcpy(d_tri, tri.data(), tri.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy");
    hipLaunchKernelGGL(smac::k_mesh_to_sdf, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, (const double*)d_tri, nf, lower[0],
                       lower[1], lower[2], res[0], res[1], res[2], dx, d_sdf, d_nrm);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "k_mesh_to_sdf launch");
    if ((e = hipMemcpy(sdf_out, d_sdf, total * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "hipMemcpy");
    if ((e = hipMemcpy(normal_out, d_nrm, total * 3 * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "hipMemcpy");
    hipFree(d_tri); hipFree(d_sdf); hipFree(d_nrm);
    return SMAC_OK;
}

int smac_create(const smac_config* cfg, smac_handle* out) {
    if (!cfg || !out) { g_create_error = "null argument"; return SMAC_ERR_INVALID; }
    if (cfg->abi_version != SMAC_ABI_VERSION) { g_create_error = "abi_version mismatch"; return SMAC_ERR_INVALID; }
    if (cfg->precision != 32 && cfg->precision != 64) { g_create_error = "precision must be 32 or 64"; return SMAC_ERR_INVALID; }
    int ndev = smac_device_count();
    if (ndev <= 0) { g_create_error = "no HIP device visible: libsoftmac_hip has no CPU fallback"; return SMAC_ERR_NOGPU; }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "device ordinal out of range"; return SMAC_ERR_INVALID; }
    ISim* s = cfg->precision == 64 ? (ISim*)new Sim<double>() : (ISim*)new Sim<float>();
    int rc = s->init(*cfg);
    if (rc != SMAC_OK) { g_create_error = s->err; delete s; return rc; }
    *out = new smac_sim{s};
    return SMAC_OK;
}
int smac_destroy(smac_handle h) {
    if (!h) return SMAC_ERR_INVALID;
    delete h->impl;
    delete h;
    return SMAC_OK;
}
#define FWD(call) (h ? h->impl->call : (int)SMAC_ERR_INVALID)
int smac_sync(smac_handle h) { return FWD(sync()); }
int smac_reset(smac_handle h, const double* state, int cols) { return FWD(reset(state, cols)); }
int smac_set_frame(smac_handle h, int f, const double* x, const double* v, const double* F, const double* C) { return FWD(set_frame(f, x, v, F, C)); }
int smac_get_state(smac_handle h, int f, double* state24) { return FWD(get_state24(f, state24)); }
int smac_get_frame(smac_handle h, int f, double* x, double* v, double* F, double* C) { return FWD(get_frame(f, x, v, F, C)); }
int smac_copy_frame(smac_handle h, int src, int dst) { return FWD(copy_frame(src, dst)); }
int smac_get_grad(smac_handle h, int f, double* gx, double* gv, double* gF, double* gC) { return FWD(get_grad(f, gx, gv, gF, gC)); }
int smac_add_grad(smac_handle h, int f, const double* gx, const double* gv, const double* gF, const double* gC) { return FWD(add_grad(f, gx, gv, gF, gC)); }
int smac_add_grad_device(smac_handle h, int f, const double* gx, const double* gv, const double* gF, const double* gC) { return FWD(add_grad_device(f, gx, gv, gF, gC)); }
int smac_clear_grads(smac_handle h) { return FWD(clear_grads()); }
int smac_carry_grad(smac_handle h, int src, int dst) { return FWD(carry_grad(src, dst)); }
int smac_set_control_idx(smac_handle h, const int32_t* idx) { return FWD(set_control_idx(idx)); }
int smac_set_material_ids(smac_handle h, const int32_t* ids) { return FWD(set_material_ids(ids)); }
int smac_set_action(smac_handle h, const double* action) { return FWD(set_action_v(action)); }
int smac_set_segment(smac_handle h, int n_live, int frame_shift) { return FWD(set_segment(n_live, frame_shift)); }
int smac_compute_grid_m(smac_handle h, int f, double* grid_m) { return FWD(compute_grid_m(f, grid_m)); }
int smac_substep(smac_handle h, int f, const double* action) { return FWD(substep(f, action)); }
int smac_substep_grad(smac_handle h, int f, const double* action, const double* ext_f_grad, double* action_grad_out) {
    return FWD(substep_grad(f, action, ext_f_grad, action_grad_out));
}
int smac_substeps(smac_handle h, int f0, int count) {
    if (!h) return SMAC_ERR_INVALID;
    for (int i = 0; i < count; ++i) {
        h->impl->hint_forward_next(i + 1 < count ? f0 + i + 1 : -1);   // lets substep f0 + i run the next substep's P2G inside its G2P launch (k_g2p_p2g)
        int rc = h->impl->substep(f0 + i, nullptr);
        h->impl->hint_forward_next(-1);
        if (rc) return rc;
    }
    return SMAC_OK;
}
int smac_substeps_grad(smac_handle h, int f0, int count, const double* ext_f_grad) {
    if (!h) return SMAC_ERR_INVALID;
    for (int i = count - 1; i >= 0; --i) {
        h->impl->hint_backward_next(i > 0 ? f0 + i - 1 : -1);       // lets substep f0 + i take the G2P adjoint of the substep before it along (k_p2g_g2p_grad)
        int rc = h->impl->substep_grad(f0 + i, nullptr, i == count - 1 ? ext_f_grad : nullptr, nullptr);
        h->impl->hint_backward_next(-1);
        if (rc) return rc;
    }
    return SMAC_OK;
}
int smac_substeps_action(smac_handle h, int f0, int count, const double* action) {
    if (!h) return SMAC_ERR_INVALID;
    if (action) { int rc = h->impl->set_action_v(action); if (rc) return rc; }
    return smac_substeps(h, f0, count);
}
int smac_substeps_grad_action(smac_handle h, int f0, int count, const double* action, const double* ext_f_grad, double* action_grad_sum) {
    if (!h) return SMAC_ERR_INVALID;
    int rc;
    if (action && (rc = h->impl->set_action_v(action))) return rc;      // zeroes action.grad; the sweep then accumulates the window's sum on the device
    if ((rc = smac_substeps_grad(h, f0, count, ext_f_grad))) return rc;
    if (action_grad_sum) return h->impl->get_action_grad(action_grad_sum);
    return SMAC_OK;
}
int smac_prim_upload_sdf(smac_handle h, int prim, const double* sdf, const double* normal, const int32_t res[3],
                         const double lower[3], const double upper[3], double sdf_dx) {
    return FWD(prim_upload_sdf(prim, sdf, normal, res, lower, upper, sdf_dx));
}
int smac_prim_set_params(smac_handle h, int prim, double friction, double softness, int contact_enabled) {
    return FWD(prim_set_params(prim, friction, softness, contact_enabled));
}
int smac_prim_set_state(smac_handle h, int prim, int f0, int f1, const double s13[13]) { return FWD(prim_set_state(prim, f0, f1, s13)); }
int smac_prim_get_state(smac_handle h, int prim, int f, double s13[13]) { return FWD(prim_get_state(prim, f, s13)); }
int smac_prim_get_state_grad(smac_handle h, int prim, int f0, int f1, double g13[13]) { return FWD(prim_get_state_grad(prim, f0, f1, g13)); }
int smac_prim_set_states(smac_handle h, int prim, int f0, int f1, const double* s13) { return FWD(prim_set_states(prim, f0, f1, s13)); }
int smac_prim_get_state_grads(smac_handle h, int prim, int f0, int f1, double* g13) { return FWD(prim_get_state_grads(prim, f0, f1, g13)); }
int smac_prim_add_state_grad(smac_handle h, int prim, int f, const double g13[13]) { return FWD(prim_add_state_grad(prim, f, g13)); }
int smac_prim_forward_kinematics(smac_handle h, int prim, int f) { return FWD(prim_fk(prim, f)); }
int smac_prim_forward_kinematics_grad(smac_handle h, int prim, int f) { return FWD(prim_fk_grad(prim, f)); }
int smac_prim_get_ext_f(smac_handle h, int prim, double ext_f[6]) { return FWD(prim_get_ext_f(prim, ext_f)); }
int smac_prim_clear_ext_f(smac_handle h, int prim) { return FWD(prim_clear_ext_f(prim)); }
int smac_prim_set_action(smac_handle h, int prim, int s, int n, const double a6[6]) { return FWD(prim_set_action(prim, s, n, a6)); }
int smac_prim_get_action_grad(smac_handle h, int prim, int s, int n, double g6[6]) { return FWD(prim_get_action_grad(prim, s, n, g6)); }
int smac_prim_get_action_grads(smac_handle h, int prim, int s0, int s1, int n, double* g6) { return FWD(prim_get_action_grads(prim, s0, s1, n, g6)); }
int smac_prim_reset(smac_handle h, int prim) { return FWD(prim_reset(prim)); }
int smac_timer_start(smac_handle h) { return FWD(timer_start()); }
int smac_timer_stop(smac_handle h, double* ms) { return FWD(timer_stop(ms)); }
int smac_profile_enable(smac_handle h, int on) { return FWD(profile_enable(on)); }
int smac_profile_reset(smac_handle h) { return FWD(profile_reset()); }
int smac_profile_count(smac_handle h) { return h ? (int)K_COUNT : (int)SMAC_ERR_INVALID; }
int smac_profile_get(smac_handle h, int i, char* name, int name_cap, double* total_ms, int64_t* launches) {
    return FWD(profile_get(i, name, name_cap, total_ms, launches));
}
int smac_count_active_cells(smac_handle h, int f, int64_t* cells) { return FWD(count_active_cells(f, cells)); }
int smac_contact_counts(smac_handle h, int32_t* nhits, int32_t* nchunks_hit) { return FWD(contact_counts(nhits, nchunks_hit)); }
int smac_set_param(smac_handle h, const char* name, double value) { return FWD(set_param(name, value)); }
int smac_get_param(smac_handle h, const char* name, double* value) { return FWD(get_param(name, value)); }
int smac_cloth_create(smac_handle h, int n_vertices, int n_faces, const int32_t* faces, int n_neighbors, const int32_t* neighbor_faces,
                      const int8_t* neighbor_dir, double friction, double softness, double cloth_force_scale, int sticky, double mpm_scale) {
    return FWD(cloth_create(n_vertices, n_faces, faces, n_neighbors, neighbor_faces, neighbor_dir, friction, softness, cloth_force_scale, sticky, mpm_scale));
}
int smac_cloth_set_state(smac_handle h, int f_begin, int f_end, const double* pos, const double* vel) { return FWD(cloth_set_state(f_begin, f_end, pos, vel)); }
int smac_cloth_get_state(smac_handle h, int f, double* pos, double* vel) { return FWD(cloth_get_state(f, pos, vel, 0)); }
int smac_cloth_get_state_grad(smac_handle h, int f, double* pos_grad, double* vel_grad) { return FWD(cloth_get_state(f, pos_grad, vel_grad, 1)); }
int smac_cloth_get_ext_f(smac_handle h, double* ext_f) { return FWD(cloth_ext_f(0, ext_f)); }
int smac_cloth_clear_ext_f(smac_handle h) { return FWD(cloth_ext_f(1, nullptr)); }
int smac_cloth_set_ext_f_grad(smac_handle h, const double* ext_f_grad) { return FWD(cloth_ext_f(2, const_cast<double*>(ext_f_grad))); }
int smac_cloth_contact_pair(smac_handle h, int f) { return FWD(cloth_contact(0, f)); }
int smac_cloth_backup_contact_pair(smac_handle h, int f) { return FWD(cloth_contact(1, f)); }
int smac_cloth_trace_penetration(smac_handle h, int f, int after_cloth) { return FWD(cloth_contact(after_cloth ? 3 : 2, f)); }
int smac_cloth_get_contact(smac_handle h, int f, int32_t* contact_id, int8_t* penetration) { return FWD(cloth_get_contact(f, contact_id, penetration)); }
int smac_cloth_set_contact(smac_handle h, int f, const int32_t* contact_id, const int8_t* penetration) { return FWD(cloth_set_contact(f, contact_id, penetration)); }
int smac_cloth_check_penetration(smac_handle h, int f, int32_t* total, int32_t* warnings) { return FWD(cloth_check_penetration(f, total, warnings)); }
int smac_loss_set_target(smac_handle h, const double* target, int m) { return FWD(loss_set_target(target, m)); }
int smac_loss_chamfer(smac_handle h, int f, double weight, int add_grad, double* loss_out) { return FWD(loss_chamfer(f, weight, add_grad, loss_out)); }
int smac_loss_min_dist(smac_handle h, int f, int id_begin, int id_end, const double center[3], double offset, double weight, int add_grad,
                       double out4[4]) {
    return FWD(loss_min_dist(f, id_begin, id_end, center, offset, weight, add_grad, out4));
}
int smac_grid_device_ptr(smac_handle h, const char* field, void** dev_ptr, int64_t* n_scalars, int32_t* scalar_bytes) {
    return FWD(grid_ptr(field, dev_ptr, n_scalars, scalar_bytes));
}
int smac_stream_handle(smac_handle h, void** hip_stream) { return FWD(stream_handle(hip_stream)); }
int smac_set_stream(smac_handle h, void* hip_stream) { return FWD(set_stream(hip_stream)); }
int smac_substep_phase(smac_handle h, int f, int phase) { return FWD(substep_phase_v(f, phase)); }
int smac_substep_grad_phase(smac_handle h, int f, const double* ext_f_grad, int phase) { return FWD(substep_grad_phase_v(f, ext_f_grad, phase)); }
int smac_comm_unique_id(char id128[128]) {
    if (!id128) { g_create_error = "null argument"; return SMAC_ERR_INVALID; }
    if (getenv("SMAC_COMM_STUB") && atoi(getenv("SMAC_COMM_STUB")) == 2) {      // the IPC test transport: the id names its shared-memory segment
        std::string e;
        if (!IpcLink::make_id(id128, e)) { g_create_error = e; return SMAC_ERR_INVALID; }
        return SMAC_OK;
    }
    Rccl& L = Rccl::get();
    if (!L.load()) { g_create_error = L.err; return SMAC_ERR_INVALID; }
    ncclUniqueId id;
    ncclResult_t r = L.GetUniqueId(&id);
    if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId failed: ") + L.GetErrorString(r); return SMAC_ERR_HIP; }
    memcpy(id128, &id, 128);
    return SMAC_OK;
}
int smac_comm_init(smac_handle h, const char id128[128], int rank, int world) { return FWD(comm_init(id128, rank, world)); }
int smac_comm_slab(smac_handle h, int left_plane0, int right_plane0, int nplanes, int contact_left, int contact_right, int base_lo, int base_hi, int self_loop) {
    return FWD(comm_slab(left_plane0, right_plane0, nplanes, contact_left, contact_right, base_lo, base_hi, self_loop));
}
int smac_substeps_slab(smac_handle h, int f0, int count) { return FWD(substeps_slab(f0, count)); }
int smac_substeps_slab_grad(smac_handle h, int f0, int count, const double* ext_f_grad) { return FWD(substeps_slab_grad(f0, count, ext_f_grad)); }
int smac_comm_allreduce_ext_f(smac_handle h, double* total_out, int clear) { return FWD(comm_allreduce_ext_f(total_out, clear)); }
int smac_comm_allreduce_prim_grad(smac_handle h, int f_begin, int f_end) { return FWD(comm_allreduce_prim_grad(f_begin, f_end)); }
int smac_comm_destroy(smac_handle h) { return FWD(comm_destroy()); }
int smac_comm_abort(smac_handle h) { return FWD(comm_abort()); }
int smac_migrate(smac_handle h, int f, int base_lo, int base_hi, int32_t out3[3]) { return FWD(migrate(f, base_lo, base_hi, out3)); }
int smac_migrate_grad(smac_handle h) { return FWD(migrate_grad()); }
int smac_set_ids(smac_handle h, const int64_t* ids) { return FWD(set_ids(ids)); }
int smac_get_ids(smac_handle h, int64_t* ids) { return FWD(get_ids(ids)); }
int smac_halo_pack(smac_handle h, const char* field, int plane0, int nplanes, void* dev_out, int minus_mixed) {
    return FWD(halo_pack(field, plane0, nplanes, dev_out, minus_mixed));
}
int smac_halo_unpack_add(smac_handle h, const char* field, int plane0, int nplanes, const void* dev_in) {
    return FWD(halo_unpack_add(field, plane0, nplanes, dev_in));
}

}  // extern "C"
