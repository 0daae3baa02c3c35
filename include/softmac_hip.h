/* softmac_hip.h - C ABI of the MI355X-native SoftMAC MPM substep engine (libsoftmac_hip.so).
 *
 * The reference exposes this path as a Python class API, not an FFI:
 *   MPMSimulator            /root/reference/softmac/engine/mpm_simulator.py:16-618
 *   Primitive / Mesh        /root/reference/softmac/engine/primitive/primitive_base.py:8-336, mesh.py:18-113
 * Each entry point below names the reference method(s) it replaces.  The Python mirror of those
 * classes (softmac_amd/engine/) binds this library with ctypes; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: opaque handle, pointers + sizes, no torch / HIP types in any signature;
 *   - every function returns 0 on success, a negative smac_status otherwise; the message is
 *     available from smac_last_error(handle) (or smac_last_error(NULL) for smac_create failures);
 *   - host arrays are caller-owned, C-contiguous float64 (the reference's public dtype,
 *     mpm_simulator.py:482-485), copied at the boundary; no pointer is retained after return;
 *   - a NULL output/input pointer means "skip this field";
 *   - all device work of a handle is queued on the handle's own HIP stream; only the get_* calls,
 *     smac_sync and the timer/profile readers block;
 *   - not re-entrant per handle; distinct handles are independent.
 */
#ifndef SOFTMAC_HIP_H
#define SOFTMAC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMAC_ABI_VERSION 2
#define SMAC_MAX_PRIMS 4

typedef struct smac_sim* smac_handle;

enum smac_status {
    SMAC_OK = 0,
    SMAC_ERR_INVALID = -1,   /* bad argument / bad frame index / wrong state */
    SMAC_ERR_HIP = -2,       /* HIP runtime error (message has the hipError string) */
    SMAC_ERR_NOMEM = -3,
    SMAC_ERR_NOGPU = -4      /* no HIP device: the product has no CPU fallback */
};

/* mpm_simulator.py:4-13 */
enum { SMAC_MODEL_COROTATED = 0, SMAC_MODEL_NEOHOOKEAN = 1 };
enum { SMAC_MAT_PLASTIC = 0, SMAC_MAT_ELASTIC = 1, SMAC_MAT_LIQUID = 2 };
enum { SMAC_CONTACT_GRID = 0, SMAC_CONTACT_PARTICLE = 1, SMAC_CONTACT_MIXED = 2 };

/* Replaces MPMSimulator.__init__ (mpm_simulator.py:17-84).  mu/lam are the already-adjusted
 * Lame parameters (:41-45); p_vol/p_mass as computed at :34-35. */
typedef struct smac_config {
    int32_t abi_version;      /* SMAC_ABI_VERSION */
    int32_t precision;        /* 32 or 64: storage + arithmetic scalar type of the device path */
    int32_t device;           /* HIP device ordinal */
    int32_t n_particles;
    int32_t n_grid;           /* int(128*quality*0.5), :26-30 */
    int32_t max_frames;       /* cfg.max_steps: frames 0..max_frames-1 are resident (:53-56) */
    int32_t grad_enabled;     /* 0: forward only (no adjoint storage) */
    int32_t substeps;         /* int(env_dt/dt), :52 - enters `life` of the forecast contact (:425) */
    int32_t ptype;            /* SMAC_MAT_* */
    int32_t material_model;   /* SMAC_MODEL_* */
    int32_t collision_type;   /* SMAC_CONTACT_* */
    int32_t n_control;        /* cfg.n_controllers (:74-77) */
    int32_t n_primitives;     /* <= SMAC_MAX_PRIMS */
    int32_t rigid_velocity_control; /* 1: substep() advances primitive poses with forward_kinematics (:329-331, 367-369) */
    int32_t sort_interval;    /* re-bin particles at most every this many substeps (0 = default 80; shortened by the library when
                                 particles are fast enough to leave their block's halo sooner); no reference counterpart */
    int32_t flags;            /* bit 0: substep_grad recomputes the forward grid like the reference (:352-359) instead of
                                 restoring the copy saved by substep (DESIGN.md "grid checkpoint");
                                 bit 1 / bit 2: slab decomposition - no wall at the low / high x end */
    int32_t adjoint_frames;   /* 0: an adjoint frame per state frame, like the reference's `needs_grad` fields (:53-56).
                                 k >= 3: ROLLING adjoint storage of k frames - substep_grad(f) only needs .grad[f] and .grad[f+1]
                                 plus the frames a loss has seeded and the backward sweep has not reached yet; a frame is released
                                 two substeps after the sweep has passed it (get_grad on it then fails).  What lets a 2000-substep
                                 episode at 1M particles fit one GPU. */
    int32_t reserved0;
    double dt;
    double mu, lam;
    double p_vol, p_mass;
    double gravity[3];
    double ground_friction;
    double yield_stress;      /* kept for API parity; the von-Mises path is dead code in the reference (:225) */
} smac_config;

const char* smac_last_error(smac_handle h);
int smac_abi_version(void);
int smac_device_count(void);   /* 0 when no GPU is visible (does not initialise a context) */

/* Asset pipeline (mesh.py:178-241, trimesh2sdf): signed distance and closest-face normal of a triangle mesh
 * sampled at lower + (i,j,k) dx, i < res[0] etc. (C order, z fastest; normal_out has a trailing 3).  Negative
 * inside.  Needs no simulation handle; errors are reported through smac_last_error(NULL). */
int smac_mesh_to_sdf(int device, const double* vertices, int nv, const int32_t* faces, int nf, const double lower[3],
                     const int32_t res[3], double dx, double* sdf_out, double* normal_out);

int smac_create(const smac_config* cfg, smac_handle* out);
int smac_destroy(smac_handle h);
int smac_sync(smac_handle h);

/* ---- particle state IO (mpm_simulator.py:448-574).  x,v: (N,3); F,C: (N,3,3); state: (N,cols). */
int smac_reset(smac_handle h, const double* state, int cols);                 /* reset :514-519, cols = 3 or 24 */
int smac_set_frame(smac_handle h, int f, const double* x, const double* v, const double* F, const double* C); /* setframe/set_x/set_v */
int smac_get_frame(smac_handle h, int f, double* x, double* v, double* F, double* C);                         /* readframe/get_x/get_v */
int smac_get_state(smac_handle h, int f, double* state24);                                                     /* get_state :541-548: (N,24) = x3 v3 F9 C9 in one transfer */
int smac_copy_frame(smac_handle h, int src, int dst);                         /* copyframe :468-479 (particles + primitives) */
int smac_get_grad(smac_handle h, int f, double* gx, double* gv, double* gF, double* gC);   /* get_grad :570-574 (+F,C) */
int smac_add_grad(smac_handle h, int f, const double* gx, const double* gv, const double* gF, const double* gC); /* loss kernels' `x.grad[f,i] +=` */
/* The same from DEVICE memory (round 4): gx ... are device pointers to C-contiguous float64 (n_particles, 3 | 3 | 9 | 9) arrays on the handle's device, in the
 * caller's particle order (NULL: none).  This is the reference's own flow - its loss kernels add to `x.grad[f, i]` on the device after the forward pass
 * (losses/loss_pour.py:130-140) - for losses the caller evaluates on the GPU (torch autograd glue): no host round trip, no synchronisation; the adds are
 * enqueued on the handle's stream (smac_stream_handle), and the caller orders the buffers' producer before this call. */
int smac_add_grad_device(smac_handle h, int f, const double* gx_dev, const double* gv_dev, const double* gF_dev, const double* gC_dev);
int smac_clear_grads(smac_handle h);                                          /* ti.ad.clear_all_gradients() */
/* Windowed episodes (checkpoint-every-K state frames with recompute; no reference counterpart: the reference keeps every frame resident).
 * smac_clear_grads, except that the particle adjoint of frame `src` survives as the adjoint of frame `dst` (with its particle order): the adjoint a
 * window's backward sweep leaves on its first frame is the seed of the window before it (softmac_amd/engine/windowed.py).  Not with rolling adjoint storage. */
int smac_carry_grad(smac_handle h, int src, int dst);
int smac_set_control_idx(smac_handle h, const int32_t* idx);                  /* set_control_idx :599-602 */
/* Two-entry material table (round 4, BASELINE config C5 "two material blocks").  The reference's mu / lam / yield_stress are per-particle fields
 * (mpm_simulator.py:47-49; soft_cloth/engine/mpm_simulator.py:47-50), filled uniformly (:86-90): a scene with two kinds of particles keeps two entries
 * and one selector per particle.  ids: n_particles entries, 0 = the material of smac_config (mu, lam, smac_set_param "yield_ratio"), 1 = entry 1
 * (smac_set_param "mu2" / "lam2" / "yield_ratio2"); indexed by the caller's particle id; NULL: one material again.  ptype / model / plasticity stay
 * per handle as in the reference.  Not with penalty contact (collision_type 1). */
int smac_set_material_ids(smac_handle h, const int32_t* ids);
/* Slab decomposition with particle MIGRATION (no reference counterpart): a handle is created with the CAPACITY
 * n_particles; a rank's live particle count changes when particles are handed to a neighbouring slab.  Frames written or
 * processed after this call hold `n_live` particles (the caller keeps frames of different segments apart: the migration point
 * occupies two consecutive frames, one per ordering); `frame_shift` = number of such duplicate frames before the current segment,
 * so that the substep phase `f % substeps` of the forecast contact (:425) follows the physical substep, not the frame index. */
int smac_set_segment(smac_handle h, int n_live, int frame_shift);
int smac_set_action(smac_handle h, const double* action);                     /* set_action :589-592: (n_control, 3); zeroes action.grad (:584-586) */
int smac_compute_grid_m(smac_handle h, int f, double* grid_m);                /* compute_grid_m_kernel :607-617, (n,n,n) out */

/* ---- the hot path (mpm_simulator.py:320-378).
 * action: (n_control,3) or NULL.  ext_f_grad: (n_primitives,6) or NULL (set_ext_f_grad :342-344).
 * action_grad_out: (n_control,3) or NULL - blocks when non-NULL (:378). */
int smac_substep(smac_handle h, int f, const double* action);
int smac_substep_grad(smac_handle h, int f, const double* action, const double* ext_f_grad, double* action_grad_out);
/* Batched forms: frames f0 .. f0+count-1 forward; f0+count-1 down to f0 backward.  One call, no
 * host round trip between substeps (replaces the python loops at taichi_env.py:101-102,128-131).
 * Both forms know which substep follows.  Forward: the G2P of substep f and the P2G of substep f+1 run in one launch where
 * the two share a binning (round 4); backward: the P2G adjoint of substep f and the G2P adjoint of substep f-1 likewise
 * (results as from count calls of smac_substep / smac_substep_grad). */
int smac_substeps(smac_handle h, int f0, int count);
int smac_substeps_grad(smac_handle h, int f0, int count, const double* ext_f_grad);
/* The same with a particle action held over the window (control_mode "mpm", taichi_env.py:101-102 / :128-133): set_action once, then the
 * substeps.  The backward form returns in action_grad_sum (n_control,3) the SUM over the window of what substep_grad returns per substep
 * (:378; TaichiEnv.step_grad adds them up, taichi_env.py:130-133): accumulated on the device, one read-back per env step.  NULLs allowed. */
int smac_substeps_action(smac_handle h, int f0, int count, const double* action);
int smac_substeps_grad_action(smac_handle h, int f0, int count, const double* action, const double* ext_f_grad, double* action_grad_sum);

/* ---- rigid primitives (primitive_base.py, mesh.py) */
int smac_prim_upload_sdf(smac_handle h, int prim, const double* sdf, const double* normal, const int32_t res[3],
                         const double lower[3], const double upper[3], double sdf_dx);          /* mesh.py:35-43 */
int smac_prim_set_params(smac_handle h, int prim, double friction, double softness, int contact_enabled); /* friction[None], softness[None], primitives_contact[i] */
int smac_prim_set_state(smac_handle h, int prim, int f_begin, int f_end, const double s13[13]);  /* set_all_states :258-260, frames [f_begin,f_end) */
int smac_prim_get_state(smac_handle h, int prim, int f, double s13[13]);                         /* get_state :248-251 (+v,w) */
int smac_prim_get_state_grad(smac_handle h, int prim, int f_begin, int f_end, double g13[13]);   /* sum of get_all_states_grad :262-265 over frames */
/* A whole trajectory in one call (prescribed primitive states of an episode / a window; their adjoints frame by frame): the per-frame forms above cost
 * a host sync each.  s13 / g13: (f_end - f_begin) x 13, frame-major. */
int smac_prim_get_action_grads(smac_handle h, int prim, int s_begin, int s_end, int n, double* g6);   /* get_action_grad of the env steps [s_begin, s_end) in one launch and one transfer: (s_end - s_begin) x 6 */
int smac_prim_set_states(smac_handle h, int prim, int f_begin, int f_end, const double* s13);
int smac_prim_get_state_grads(smac_handle h, int prim, int f_begin, int f_end, double* g13);
int smac_prim_add_state_grad(smac_handle h, int prim, int f, const double g13[13]);              /* loss kernels' position/v/w .grad[f] += */
int smac_prim_forward_kinematics(smac_handle h, int prim, int f);                                /* forward_kinematics :280-283 */
int smac_prim_forward_kinematics_grad(smac_handle h, int prim, int f);                           /* forward_kinematics.grad */
int smac_prim_get_ext_f(smac_handle h, int prim, double ext_f[6]);                               /* ext_f.to_numpy() */
int smac_prim_clear_ext_f(smac_handle h, int prim);                                              /* clear_ext_f :183-187 */
int smac_prim_set_action(smac_handle h, int prim, int s, int n, const double a6[6]);             /* set_action :311-313 */
int smac_prim_get_action_grad(smac_handle h, int prim, int s, int n, double g6[6]);              /* get_action_grad :315-319 */
int smac_prim_reset(smac_handle h, int prim);                                                    /* reset :271-275 */

/* ---- measurement (bench.py).  HIP events on the handle's stream. */
int smac_timer_start(smac_handle h);
int smac_timer_stop(smac_handle h, double* elapsed_ms);          /* blocks */
int smac_profile_enable(smac_handle h, int on);                  /* per-kernel event pairs */
int smac_profile_reset(smac_handle h);
int smac_profile_count(smac_handle h);                           /* number of kernel classes */
int smac_profile_get(smac_handle h, int i, char* name, int name_cap, double* total_ms, int64_t* launches); /* blocks */
int smac_count_active_cells(smac_handle h, int f, int64_t* cells); /* cells with grid_m > 0 after P2G of frame f (G_t of SURVEY 8d) */
/* ---- losses on the device (losses/loss_pour.py:44-70, loss_grip.py:45-68: chamfer_closest + compute_chamfer_loss_kernel
 * and the tape's adjoint of the latter).  set_target uploads the (m,3) target cloud once; chamfer returns
 *   sum_i min_j |x_i - t_j|^2 + sum_j min_i |x_i - t_j|^2   over frame f (unweighted), and with add_grad != 0 also does
 *   x.grad[f] += weight * d(chamfer)/dx  - what `with ti.ad.Tape(loss)` leaves in x.grad[f] for a loss weight `weight`. */
int smac_loss_set_target(smac_handle h, const double* target, int m);
int smac_loss_chamfer(smac_handle h, int f, double weight, int add_grad, double* loss_out);
/* loss_door.py:49-61 / loss_transport.py:58-76: value = min over the particles with ids in [id_begin, id_end) of
 * max(|x_i - center|^2 - offset, 0) at frame f.  out4 = {value, d(weight value^2)/d center}; with add_grad != 0 the seed
 * d(weight value^2)/dx of the winning particle is added to x.grad[f] (Taichi's atomic_min routes the adjoint to the winner). */
int smac_loss_min_dist(smac_handle h, int f, int id_begin, int id_end, const double center[3], double offset, double weight, int add_grad,
                       double out4[4]);
int smac_contact_counts(smac_handle h, int32_t* nhits, int32_t* nchunks_hit); /* particles inside a contact band after the last forward substep (nchunks_hit: retired, 0) */

/* ---- material options of the soft_cloth variant of the simulator (soft_cloth/engine/mpm_simulator.py) ----
 * "plasticity": 0 = singular values clipped to [1-2e-3, 1+3e-3] (softmac mpm_simulator.py:226-229, default),
 *               1 = von-Mises return mapping (soft_cloth mpm_simulator.py:172-188, :232);
 * "yield_ratio": yield_stress / (2 mu) of that return mapping (:182);
 * "mass_eps":   a grid node has a velocity when its mass exceeds this (1e-10, mpm_simulator.py:286/399; the Python mirror rescales
 *               it by mpm_scale^-2 because the particle kernels work on the unit domain);
 * "cloth_pairs_flat": 1 = smac_cloth_contact_pair tests every particle against every face also on sorted frames (default 0: faces are
 *               culled per chunk first; same result) */
int smac_set_param(smac_handle h, const char* name, double value);
/* reads a parameter back; also "drift_repairs": how many times an epoch was recomputed because a particle out-ran its binning (no reference
 * counterpart: the reference's dense grid has no binning); "contact_skips": backward substeps that launched no contact adjoint because the hit list
 * filed with the frame's grid checkpoint was empty; "max_hits": the longest contact hit list filed so far (synchronises; -1: none filed yet - the
 * library sizes the two hit-list launches from these counts); "chunks": work items (<= 256 particles of one grid block each, one workgroup per item)
 * of the binning in use */
int smac_get_param(smac_handle h, const char* name, double* value);

/* ---- cloth primitive: replaces soft_cloth/engine/primitive/primitive_cloth.py (Primitive_Cloth) and the contact bookkeeping of
 * soft_cloth/engine/mpm_simulator.py:447-561.  One per handle, instead of SDF primitives.  Vertex positions / velocities /
 * ext_f and their adjoints are float64 (n_vertices, 3) arrays in PHYSICAL units (the domain [0, mpm_scale)^3); the sheet moves
 * kinematically (set per frame by the caller, as cloth_simulator.py:80-81 does with DiffClothAI's result). ---- */
int smac_cloth_create(smac_handle h, int n_vertices, int n_faces, const int32_t* faces /* (n_faces,3) */, int n_neighbors,
                      const int32_t* neighbor_faces /* (n_faces,n_neighbors), process_faces.py */, const int8_t* neighbor_dir,
                      double friction, double softness, double cloth_force_scale, int sticky, double mpm_scale);   /* :30-69, initialize :367-376 */
int smac_cloth_set_state(smac_handle h, int f_begin, int f_end, const double* pos, const double* vel);   /* set_all_states :348-354 for frames [f_begin, f_end) */
int smac_cloth_get_state(smac_handle h, int f, double* pos, double* vel);                                /* get_all_states :320-324 */
int smac_cloth_get_state_grad(smac_handle h, int f, double* pos_grad, double* vel_grad);                 /* get_all_states_grad :334-338 */
int smac_cloth_get_ext_f(smac_handle h, double* ext_f);                                                  /* ext_f.to_numpy() */
int smac_cloth_clear_ext_f(smac_handle h);                                                               /* clear_ext_f :285-290 (value and grad) */
int smac_cloth_set_ext_f_grad(smac_handle h, const double* ext_f_grad);                                  /* set_ext_f_grad :292-296 */
int smac_cloth_contact_pair(smac_handle h, int f);                 /* get_contact_pair, mpm_simulator.py:447-469 */
int smac_cloth_backup_contact_pair(smac_handle h, int f);          /* backup_contact_pair :471-482 */
int smac_cloth_trace_penetration(smac_handle h, int f, int after_cloth);   /* trace_penetration_after_mpm :484-518 (0) / _after_cloth :520-553 (1) */
int smac_cloth_get_contact(smac_handle h, int f, int32_t* contact_id, int8_t* penetration);   /* the two extra columns of readframe :567-578; either may be NULL */
int smac_cloth_set_contact(smac_handle h, int f, const int32_t* contact_id, const int8_t* penetration);   /* reset_all_kernel :642-643 / reset_kernel :629 */
int smac_cloth_check_penetration(smac_handle h, int f, int32_t* total, int32_t* warnings);   /* check_penetration :555-561; warnings (may be NULL): particles the
                                                                                                 tracing skipped because their previous face was not a listed neighbour (:509, :544) */

/* ---- raw device views for the multi-GPU halo exchange (softmac_amd/parallel.py wraps them as
 * torch tensors for torch.distributed/RCCL; no reference counterpart - SURVEY 8e). */
int smac_grid_device_ptr(smac_handle h, const char* field, void** dev_ptr, int64_t* n_scalars, int32_t* scalar_bytes);
int smac_stream_handle(smac_handle h, void** hip_stream);
int smac_set_stream(smac_handle h, void* hip_stream);   /* run on the caller's stream (e.g. torch's current stream, so RCCL ops order with the kernels) */

/* ---- slab decomposition over several GPUs (softmac_amd/parallel.py drives it with torch.distributed).
 * config.flags bit 1 / bit 2: a neighbour slab sits at the low / high x end (no wall there).
 * smac_substep_phase: phase 0 = clear + p2g + slab reduction  -> then sum "grid_in" halo planes with the neighbours
 *                     phase 1 = grid_op + contact            -> then sum the contact corrections of "grid_out" (minus_mixed = 1)
 *                     phase 2 = grid checkpoint + g2p
 * smac_substep_grad_phase: phase 0 = restore forward grid + g2p.grad -> then sum "grid_out.grad" halo planes
 *                          phase 1 = contact adjoint                -> then sum "grid_mixed.grad" halo planes
 *                          phase 2 = grid_op.grad + p2g.grad
 * Halo buffers are dense (nplanes, n, n) arrays of 4-scalar records in the handle's precision, in DEVICE memory. */
int smac_substep_phase(smac_handle h, int f, int phase);
int smac_substep_grad_phase(smac_handle h, int f, const double* ext_f_grad, int phase);
int smac_halo_pack(smac_handle h, const char* field, int plane0, int nplanes, void* dev_out, int minus_mixed);
int smac_halo_unpack_add(smac_handle h, const char* field, int plane0, int nplanes, const void* dev_in);

/* ---- the same slab loop INSIDE the library, on RCCL (round 3; SURVEY 8e "Collective": ncclSend / ncclRecv inside ncclGroupStart/End with the two
 * slab neighbours only, on a communication stream of the handle's own, ordered against the kernels with events; a small ncclAllReduce for the
 * primitives' wrench sums and state adjoints, once per env step where the reference consumes them - rigid_simulator.py:92-93, 203-208).
 * RCCL is resolved at run time (dlopen librccl.so.1; SMAC_RCCL_LIB overrides): a single-GPU user never loads it.
 *   smac_comm_unique_id     rank 0 makes the 128-byte id (ncclGetUniqueId); the caller hands it to every rank (bench.py: torch.distributed gloo)
 *   smac_comm_init          ncclCommInitRank on the handle's device
 *   smac_comm_slab          this rank's geometry: first shared x-plane with the left / right neighbour (this rank's grid indexing), planes per side,
 *                           whether a contact primitive can reach that side's planes (both neighbours must pass the same flag), the range of
 *                           stencil bases (x) this rank's planes cover (a particle outside it raises an error at the next sync instead of losing
 *                           its deposits); self_loop = 1 (world 1 only): left = right = this rank on periodic planes - the whole RCCL path on ONE GPU
 *   smac_substeps_slab[_grad]   count substeps forward / backward with the 2 + 2 exchanges per substep pair, no host round trip in between; the fused particle
 *                           launches of smac_substeps[_grad] are taken here too (they cross no exchange), the grid passes stay in pieces around the exchanges
 *   smac_comm_allreduce_ext_f   SUM of ext_f over the ranks in place; total_out (n_primitives, 6) or NULL; clear = the reference's clear_ext_f after the read
 *   smac_comm_allreduce_prim_grad   SUM of the primitive-state adjoints of frames [f_begin, f_end) over the ranks, in place */
int smac_comm_unique_id(char id128[128]);
int smac_comm_init(smac_handle h, const char id128[128], int rank, int world);
int smac_comm_slab(smac_handle h, int left_plane0, int right_plane0, int nplanes, int contact_left, int contact_right, int base_lo, int base_hi, int self_loop);
int smac_substeps_slab(smac_handle h, int f0, int count);
int smac_substeps_slab_grad(smac_handle h, int f0, int count, const double* ext_f_grad);
int smac_comm_allreduce_ext_f(smac_handle h, double* total_out, int clear);
int smac_comm_allreduce_prim_grad(smac_handle h, int f_begin, int f_end);
int smac_comm_destroy(smac_handle h);
/* Failure inside the collective loop.  A rank whose smac_substeps_slab[_grad] / smac_migrate[_grad] fails while it holds a communicator of world > 1
 * aborts that communicator itself (ncclCommAbort) before it returns the error: its neighbours have the matching receive enqueued and nothing will
 * ever answer it, so the caller must END THE PROCESS with a non-zero status (the launcher then stops the peers) or, with a control plane of its own,
 * tell the other ranks, which call smac_comm_abort - ncclCommAbort without the stream sync of smac_comm_destroy, which would wait for a receive
 * that cannot complete.  Nothing is retried in-process.  (No reference counterpart: the reference is single-device.) */
int smac_comm_abort(smac_handle h);
/* Particle migration between slabs on the device (SURVEY 8e "Particle migration"; round 2 went through the host with get_state / set_state).
 * smac_migrate: the particles of frame f whose stencil base (x) left [base_lo, base_hi) go to the neighbour on that side (rows and global ids over
 * RCCL, neighbour-only); frame f + 1 starts the next segment = kept particles, arrivals from the left, arrivals from the right (the caller goes on
 * from frame f + 1; smac_set_segment's bookkeeping is done here).  out3 = {live particles now, sent away, received}.  Called at a re-sort /
 * env-step boundary, never inside a substep.  smac_migrate_grad undoes the most recent one in the backward sweep: the adjoint of frame f + 1 is
 * added into frame f's, across the slab boundary for the particles that crossed it.  smac_set_ids / smac_get_ids: the global particle ids of the
 * current segment in its caller order (default 0 .. n_particles - 1); they travel with the particles. */
int smac_migrate(smac_handle h, int f, int base_lo, int base_hi, int32_t out3[3]);
int smac_migrate_grad(smac_handle h);
int smac_set_ids(smac_handle h, const int64_t* ids);
int smac_get_ids(smac_handle h, int64_t* ids);

#ifdef __cplusplus
}
#endif
#endif /* SOFTMAC_HIP_H */
