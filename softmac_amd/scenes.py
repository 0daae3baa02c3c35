"""Synthetic, seeded workloads of SURVEY.md section 8(d) (no datasets are needed or available).

S-grip (BASELINE config C3): 1,048,576 particles at 8 per cell on a 128^3 grid, plasticine
(fixed-corotated + sigma clamp), sticky floor, three gripper primitives (palm disabled, two fingers
touching the block and closing at 0.3 m/s) with forecast contact.
S-elastic (C2): 262,144 particles, 64^3, elastic, no primitives.
"""
from __future__ import annotations

import types

import numpy as np


def block_cloud(n_particles, n_grid, center, ppc=8, seed=0, v_std=0.1, C_std=1.0, F_std=5e-3):
    """Uniform random block with `ppc` particles per cell: side = (N/ppc)^(1/3) cells."""
    rng = np.random.default_rng(seed)
    side = (n_particles / ppc) ** (1.0 / 3.0) / n_grid
    lo = np.asarray(center) - side / 2
    x = lo + side * rng.random((n_particles, 3))
    v = v_std * rng.standard_normal((n_particles, 3))
    F = np.eye(3).reshape(1, 9) + F_std * rng.standard_normal((n_particles, 9))
    C = C_std * rng.standard_normal((n_particles, 9))
    return np.hstack([x, v, F, C]), lo, side


def cylinder_sdf(radius=0.05, height=0.2, dx=0.0025, margin=0.01):
    """Voxelised SDF of the gripper finger (a y-axis cylinder r=0.05, h=0.2, as built by the reference's
    assets/gripper/build_gripper_mesh.py) on the grid the reference's voxeliser would choose
    (mesh.py:170-176, 190-233: dx = min(0.01, extent/80), margin = max(3dx, 0.01), samples at cell centres,
    negative inside, normal = normal of the closest face)."""
    half = np.array([radius, height / 2, radius])
    res = np.ceil((2 * half + 2 * margin) / dx).astype(int)
    lower = -res * dx / 2.0
    ax = [np.arange(0.5, r) * dx + l for r, l in zip(res, lower)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    rho = np.sqrt(X * X + Z * Z)
    dr = rho - radius                    # signed distance to the side
    dy = np.abs(Y) - height / 2          # signed distance to the caps
    outside = np.sqrt(np.maximum(dr, 0) ** 2 + np.maximum(dy, 0) ** 2)
    inside = np.minimum(np.maximum(dr, dy), 0)
    sdf = outside + inside
    side_closer = dr > dy                # which face is closest
    nrm = np.zeros(sdf.shape + (3,))
    safe = np.maximum(rho, 1e-12)
    nrm[..., 0] = np.where(side_closer, X / safe, 0.0)
    nrm[..., 2] = np.where(side_closer, Z / safe, 0.0)
    nrm[..., 1] = np.where(side_closer, 0.0, np.sign(Y))
    lower = lower + dx / 2.0
    upper = lower + (res - 1) * dx
    return dict(sdf=sdf, normal=nrm, lower=lower, upper=upper, dx=dx, res=res)


def box_sdf(half=(0.3, 0.15, 0.075), dx=0.0075, margin=0.0225):
    """Voxelised SDF of the gripper palm box (assets/gripper/palm.obj), same sampling rule."""
    half = np.asarray(half)
    res = np.ceil((2 * half + 2 * margin) / dx).astype(int)
    lower = -res * dx / 2.0
    ax = [np.arange(0.5, r) * dx + l for r, l in zip(res, lower)]
    P = np.stack(np.meshgrid(*ax, indexing="ij"), -1)
    q = np.abs(P) - half
    sdf = np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(q.max(-1), 0)
    k = np.argmax(q, axis=-1)
    nrm = np.zeros_like(P)
    np.put_along_axis(nrm, k[..., None], np.take_along_axis(np.sign(P), k[..., None], -1), -1)
    lower = lower + dx / 2.0
    upper = lower + (res - 1) * dx
    return dict(sdf=sdf, normal=nrm, lower=lower, upper=upper, dx=dx, res=res)


def sim_namespace(**kw):
    base = dict(dim=3, dtype="float64", quality=1, yield_stress=30., ground_friction=20., gravity=(0., -9.8, 0.),
                n_particles=0, dt=1e-4, ptype=0, material_model=0, E=3e3, nu=0.2, max_steps=128, n_controllers=0,
                collision_type=2, n_grid=128, precision="float32", device=0, grad_enabled=True)
    base.update(kw)
    return types.SimpleNamespace(**base)


def table_spec(t):
    """SDF table in the layout a primitive spec carries, from either layout in use: the reference's cache dict (`position` = (lower, upper),
    `dx` a 3-vector: what `engine.primitive.voxelize.mesh_to_sdf` returns and `mesh.py:214-241` pickles) or lower / upper / scalar dx."""
    if "position" in t:
        return dict(sdf=np.asarray(t["sdf"]), normal=np.asarray(t["normal"]), lower=np.asarray(t["position"][0], dtype=np.float64),
                    upper=np.asarray(t["position"][1], dtype=np.float64), dx=float(np.asarray(t["dx"]).reshape(-1)[0]), res=np.asarray(t["res"]))
    return dict(sdf=np.asarray(t["sdf"]), normal=np.asarray(t["normal"]), lower=np.asarray(t["lower"], dtype=np.float64),
                upper=np.asarray(t["upper"], dtype=np.float64), dx=float(t["dx"]), res=np.asarray(t["res"]))


def grip_dt(n_grid):
    """dt = 1e-4 at dx = 1/128 keeps the plastic block's elastic wave (c = sqrt((lam + 2 mu) / rho) = 57.7 m/s) at c dt / dx = 0.74;
    the same Courant number at any other resolution (2e-4 at 64^3 - the reference's own demo_grip_config.py:25 - and 5e-5 at 256^3).  Round 4 kept
    1e-4 at 256^3 (c dt / dx = 1.48) and the scene diverged from frame 22 on."""
    return 1e-4 * 128.0 / n_grid


def s_grip(n_particles=1 << 20, n_grid=128, max_steps=128, precision="float32", device=0, seed=1, dt=None,
           substeps=10, x_offset=0.0, tables=None):
    """Returns (cfg namespace, env_dt, state24, primitive specs, primitive state13 at frame 0).  dt=None: `grip_dt(n_grid)`.

    tables = (palm, finger): the gripper's SDF tables as SURVEY 8(d) asks for them - the palm from the reference's own cache
    (assets/gripper/6895...c4d5), the finger voxelised from assets/gripper/finger.obj by the library's mesh -> SDF kernel (row f3); bench.py
    passes them (`gripper_tables`).  None: tables built analytically with the reference voxeliser's sampling rule (the tests' default: no
    fixture files, no GPU call before the simulator exists; the palm table equals the cache to 2e-16, the finger is a true cylinder instead
    of finger.obj's 32-sided prism)."""
    if dt is None:
        dt = grip_dt(n_grid)
    center = (0.5 + x_offset, 0.3, 0.5)
    state, lo, side = block_cloud(n_particles, n_grid, center, ppc=8, seed=seed, v_std=0.05, C_std=0.5, F_std=5e-3)
    cfg = sim_namespace(n_particles=n_particles, n_grid=n_grid, dt=dt, max_steps=max_steps, precision=precision,
                        device=device)
    palm, finger = (box_sdf(), cylinder_sdf()) if tables is None else (table_spec(tables[0]), table_spec(tables[1]))
    specs = [dict(palm, friction=0.001, softness=666.0, contact=False),     # demo_grip.py:117 [False, True, True]
             dict(finger, friction=0.001, softness=666.0, contact=True),
             dict(finger, friction=0.001, softness=666.0, contact=True)]
    ident = [1.0, 0.0, 0.0, 0.0]
    yc = center[1]
    x_l, x_r = lo[0] - 0.05 + 0.002, lo[0] + side + 0.05 - 0.002          # fingers just biting into the block
    s13 = [np.array([center[0], lo[1] + side + 0.2, 0.5] + ident + [0, 0, 0] + [0, 0, 0], dtype=np.float64),
           np.array([x_l, yc, 0.5] + ident + [0.3, 0, 0] + [0, 0, 0], dtype=np.float64),
           np.array([x_r, yc, 0.5] + ident + [-0.3, 0, 0] + [0, 0, 0], dtype=np.float64)]
    return cfg, dt * substeps, state, specs, s13


def s_elastic(n_particles=1 << 18, n_grid=64, max_steps=128, precision="float32", device=0, seed=0):
    state, lo, side = block_cloud(n_particles, n_grid, (0.5, 0.5, 0.5), ppc=8, seed=seed, v_std=0.1, C_std=1.0, F_std=0.01)
    cfg = sim_namespace(n_particles=n_particles, n_grid=n_grid, dt=2e-4, ptype=1, max_steps=max_steps,
                        precision=precision, device=device, ground_friction=1.5)
    return cfg, 2e-3, state, [], []


def s_grip_slab(rank, world, n_particles=1 << 20, n_grid=128, max_steps=128, precision="float32", device=0, seed=1, dt=None,
                substeps=10, lo=32, hi=96):
    """Weak-scaling variant of S-grip for several GPUs: one continuous bar of plasticine along x, cut into
    `world` slabs.  Rank r simulates in local coordinates on its own n_grid^3 grid and owns the particles whose
    stencil base lies in x-planes [lo, hi); its planes [hi, hi+2) are the same physical planes as the right
    neighbour's [lo, lo+2).  Same particle count, density and primitives per GPU as the single-GPU workload
    (the two fingers close on the bar's z faces)."""
    if dt is None:
        dt = grip_dt(n_grid)
    rng = np.random.default_rng(seed + 1000 * rank)
    wx = hi - lo
    side = (n_particles / 8 / wx) ** 0.5                     # y,z extent in cells at 8 particles per cell
    dx = 1.0 / n_grid
    x = np.empty((n_particles, 3))
    x[:, 0] = (lo + 0.5 + wx * rng.random(n_particles)) * dx * (1 - 1e-9)
    y0, z0 = 0.3 - side * dx / 2, 0.5 - side * dx / 2
    x[:, 1] = y0 + side * dx * rng.random(n_particles)
    x[:, 2] = z0 + side * dx * rng.random(n_particles)
    v = 0.05 * rng.standard_normal((n_particles, 3))
    F = np.eye(3).reshape(1, 9) + 5e-3 * rng.standard_normal((n_particles, 9))
    C = 0.5 * rng.standard_normal((n_particles, 9))
    state = np.hstack([x, v, F, C])
    cfg = sim_namespace(n_particles=n_particles, n_grid=n_grid, dt=dt, max_steps=max_steps, precision=precision, device=device)
    cfg.slab_flags = (2 if rank > 0 else 0) | (4 if rank < world - 1 else 0)      # smac_config.flags bits 1,2
    finger, palm = cylinder_sdf(), box_sdf()
    specs = [dict(palm, friction=0.001, softness=666.0, contact=False),
             dict(finger, friction=0.001, softness=666.0, contact=True),
             dict(finger, friction=0.001, softness=666.0, contact=True)]
    ident = [1.0, 0.0, 0.0, 0.0]
    xc = (lo + wx / 2) * dx
    s13 = [np.array([xc, y0 + side * dx + 0.2, 0.5] + ident + [0, 0, 0] + [0, 0, 0], dtype=np.float64),
           np.array([xc, 0.3, z0 - 0.05 + 0.002] + ident + [0, 0, 0.3] + [0, 0, 0], dtype=np.float64),
           np.array([xc, 0.3, z0 + side * dx + 0.05 - 0.002] + ident + [0, 0, -0.3] + [0, 0, 0], dtype=np.float64)]
    return cfg, dt * substeps, state, specs, s13, (lo, hi)


def balanced_slab_bounds(base_x, world, n_grid, min_width=2):
    """Slab boundaries from the per-plane particle prefix sum: rank r owns stencil bases lo[r] <= base.x < lo[r+1].
    Interior slabs are at least `min_width` planes wide: with a drift tolerance of t cells a slab's particles touch planes
    [lo - t, hi + 2 + t), and only NEIGHBOURING slabs exchange, so slabs r-1 and r+1 must not meet: width >= 2 + 2 t."""
    cnt = np.bincount(np.clip(base_x, 0, n_grid - 1), minlength=n_grid)
    cum = np.concatenate([[0], np.cumsum(cnt)])
    occ = np.nonzero(cnt)[0]
    first, last = int(occ[0]), int(occ[-1]) + 1
    if last - first < min_width * max(world - 2, 0) + 2 * min(world - 1, 1):
        raise ValueError(f"{world} slabs of >= {min_width} planes do not fit the {last - first} occupied planes")
    bounds = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        b = int(np.argmin(np.abs(cum - target)))
        lo_ok = first + 1 if r == 1 else bounds[-1] + min_width
        hi_ok = last - 1 - min_width * (world - 1 - r)
        bounds.append(min(max(b, lo_ok), hi_ok))
    bounds.append(n_grid)
    return bounds


def gripper_tables(fixture_dir, device=0):
    """(palm, finger) for `s_grip(tables=...)`: the palm table is the reference's cached SDF as shipped in its tree (tests/golden/palm_sdf.npz,
    made from the pickle by tools/make_fixtures.py); the finger table is computed HERE from finger.obj's vertex / face arrays
    (tests/golden/grip_scene.npz) by smac_mesh_to_sdf with the reference's sampling box (mesh.py:170-176, 190-233) - the finger's cache blob is
    one of the three missing from the reference checkout (.MISSING_LARGE_BLOBS)."""
    import os
    from .engine.primitive import voxelize
    palm = np.load(os.path.join(fixture_dir, "palm_sdf.npz"))
    grip = np.load(os.path.join(fixture_dir, "grip_scene.npz"))
    finger = voxelize.mesh_to_sdf(grip["finger_vertices"], grip["finger_faces"], device=device)
    return dict(sdf=palm["sdf"], normal=palm["normal"], lower=palm["lower"], upper=palm["upper"], dx=float(palm["dx"]), res=palm["res"]), finger


def s_grip_strong(rank, world, n_particles=1 << 20, n_grid=128, max_steps=128, precision="float32", device=0, seed=1, dt=None,
                  substeps=10, drift_tol=1, tables=None):
    """Strong-scaling form of S-grip (the metric's "1M particles / 128^3 on 1/2/4/8 GPUs"): the SAME scene as `s_grip`
    - one block, one global grid, the shared gripper primitives - cut into `world` x-slabs balanced by particle count.
    Every rank works in global coordinates on its own copy of the block-sparse grid (only its slab's blocks are active).
    Returns (cfg, env_dt, state of the owned particles, specs, s13, (left_plane0, right_plane0, nplanes), owned ids)."""
    cfg, env_dt, state, specs, s13 = s_grip(n_particles, n_grid, max_steps, precision, device, seed=seed, dt=dt, substeps=substeps, tables=tables)
    base = (state[:, 0] * n_grid - 0.5).astype(np.int64)
    bounds = balanced_slab_bounds(base, world, n_grid, min_width=2 + 2 * drift_tol)
    lo, hi = bounds[rank], bounds[rank + 1]
    own = np.nonzero((base >= lo) & (base < hi))[0]
    cfg.n_particles = len(own)
    cfg.slab_flags = (2 if rank > 0 else 0) | (4 if rank < world - 1 else 0)      # smac_config.flags bits 1,2: open x ends
    nplanes = 2 + 2 * drift_tol
    return cfg, env_dt, state[own], specs, s13, (max(lo - drift_tol, 0), min(hi - drift_tol, n_grid - nplanes), nplanes), own


def s_pour(n_particles=1 << 22, n_grid=256, max_steps=16, precision="float32", device=0, seed=2, bowl_table=None):
    """S-pour (BASELINE config C4): a column of liquid (ptype 2, E 22) at 8 particles per cell on a 256^3 grid just above /
    inside a bowl SDF; dt = 2.5e-4 (the reference's 1e-3 at dx = 1/64 scaled by the CFL ratio).  `bowl_table`: the voxelised
    bowl (dict with sdf, normal, position, dx, res as the voxeliser returns it); None -> no primitive."""
    side_cells = (n_particles / 8) ** (1.0 / 3.0)
    center = (0.5, 0.5, 0.5)
    state, lo, side = block_cloud(n_particles, n_grid, center, ppc=8, seed=seed, v_std=0.05, C_std=0.5, F_std=0.0)
    state[:, 6:15] = (np.eye(3) * 0.99).reshape(1, 9)                     # slightly compressed liquid, like the reference's pour fixture (F = 0.98 I)
    state[:, 4] -= 0.5                                                    # falling
    cfg = sim_namespace(n_particles=n_particles, n_grid=n_grid, dt=2.5e-4, max_steps=max_steps, precision=precision, device=device,
                        ptype=2, E=22.0, ground_friction=0.0)
    specs, s13 = [], []
    if bowl_table is not None:
        t = bowl_table
        specs = [dict(sdf=t["sdf"], normal=t["normal"], lower=np.asarray(t["position"][0]), upper=np.asarray(t["position"][1]),
                      dx=float(np.asarray(t["dx"]).reshape(-1)[0]), res=np.asarray(t["res"]), friction=1.0, softness=666.0, contact=True)]
        # bowl rim (0.067 above its origin) 2 mm inside the bottom of the column
        s13 = [np.array([0.5, lo[1] - 0.0668 + 0.002, 0.5, 1.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, 0, 0], dtype=np.float64)]
    return cfg, 1e-3, state, specs, s13


def tortilla_disc(rings=12, radius=1.0):
    """A flat triangulated disc in the y = 0 plane (the shape of the reference's `envs/assets/tortilla/tortilla.obj`: a centre vertex and
    concentric rings, 6 k vertices on ring k): vertices (1 + 3 rings (rings + 1), 3), consistently oriented faces (6 rings^2, 3)."""
    verts = [(0.0, 0.0, 0.0)]
    start = [0]
    for k in range(1, rings + 1):
        start.append(len(verts))
        for j in range(6 * k):
            a = 2 * np.pi * j / (6 * k)
            verts.append((radius * k / rings * np.cos(a), 0.0, radius * k / rings * np.sin(a)))
    faces = []
    for k in range(1, rings + 1):
        n_out, n_in = 6 * k, max(6 * (k - 1), 1)
        i = j = 0
        while i < n_out or j < (n_in if k > 1 else 0):
            a_out, a_in = (i + 1) / n_out, ((j + 1) / n_in if k > 1 else 2.0)
            o0, o1 = start[k] + i % n_out, start[k] + (i + 1) % n_out
            i0 = start[k - 1] + (j % n_in if k > 1 else 0)
            if a_out <= a_in or j >= n_in:
                faces.append((o0, i0, o1)); i += 1
            else:
                i1 = start[k - 1] + (j + 1) % n_in
                faces.append((o0, i0, i1)); j += 1
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int32)


def s_taco(n_particles=1 << 20, n_grid=128, max_steps=64, precision="float32", device=0, seed=3, rings=12):
    """Soft <-> cloth workload shaped after the reference's taco demo (soft_cloth/config/demo_taco_config.py): mpm_scale 5, von-Mises
    plasticine (E 5000, yield stress 60), gravity (0,-5,0), dt 2e-4 / env_dt 2e-3, a sticky sheet = triangulated disc of radius 1.5 at
    y = 2.0; the particles are a cylinder of 8 per cell resting on it.  Returns (cfg, env_dt, scale, state (N,3+21), vertices, faces, prim)."""
    rng = np.random.default_rng(seed)
    scale = 5.0
    dx = scale / n_grid
    r = 40.0 * n_grid / 128 * dx
    h = n_particles / 8.0 * dx ** 3 / (np.pi * r * r)
    rr = r * np.sqrt(rng.random(n_particles))
    th = 2 * np.pi * rng.random(n_particles)
    x = np.stack([2.5 + rr * np.cos(th), 2.0 + 0.1 * dx + h * rng.random(n_particles), 2.5 + rr * np.sin(th)], 1)
    v = np.tile([0.0, -0.3, 0.0], (n_particles, 1)) + 0.02 * rng.standard_normal((n_particles, 3))
    F = np.eye(3).reshape(1, 9) + 5e-3 * rng.standard_normal((n_particles, 9))
    C = 0.3 * rng.standard_normal((n_particles, 9))
    V, Fc = tortilla_disc(rings, 1.5)
    V = V + np.array([2.5, 2.0, 2.5])
    cfg = sim_namespace(n_particles=n_particles, n_grid=n_grid, dt=2e-4, E=5000.0, nu=0.2, ptype=0, material_model=0, gravity=(0.0, -5.0, 0.0),
                        ground_friction=0.0, collision_type=2, n_controllers=0, max_steps=max_steps, precision=precision, device=device,
                        yield_stress=60.0)
    prim = dict(friction=1.0, softness=666.0, cloth_force_scale=1.0, mpm_force_scale=1.0, sticky=True)
    return cfg, 2e-3, scale, np.hstack([x, v, F, C]), V, Fc, prim


def s_mixed(n_particles=1 << 20, n_grid=128, max_steps=64, precision="float32", device=0, seed=3, rings=12, palm=None):
    """S-mixed (BASELINE config C5 "mixed soft-rigid-cloth"): `s_taco` on the unit domain (mpm_scale 1: every length / 5, E and the yield
    stress / 25, gravity / 5 - the same dimensionless problem) with TWO material blocks (x > 0.5: half the stiffness, a lower yield stress), the sticky
    sheet under the cylinder and ONE rigid SDF primitive - a box (`palm`: an SDF table spec, e.g. the reference's cached gripper palm) pressed into the
    cylinder's top at 0.2 m/s.  Returns (cfg, env_dt, state (N, 24), vertices, faces, sheet cfg, rigid spec, rigid state13, mat_id, material-2 dict)."""
    rng = np.random.default_rng(seed)
    dx = 1.0 / n_grid
    r = 40.0 * n_grid / 128 * dx
    h = n_particles / 8.0 * dx ** 3 / (np.pi * r * r)
    rr = r * np.sqrt(rng.random(n_particles))
    th = 2 * np.pi * rng.random(n_particles)
    x = np.stack([0.5 + rr * np.cos(th), 0.4 + 0.1 * dx + h * rng.random(n_particles), 0.5 + rr * np.sin(th)], 1)
    v = np.tile([0.0, -0.06, 0.0], (n_particles, 1)) + 0.004 * rng.standard_normal((n_particles, 3))
    F = np.eye(3).reshape(1, 9) + 5e-3 * rng.standard_normal((n_particles, 9))
    C = 0.3 * rng.standard_normal((n_particles, 9))
    V, Fc = tortilla_disc(rings, 0.3)
    V = V + np.array([0.5, 0.4, 0.5])
    cfg = sim_namespace(n_particles=n_particles, n_grid=n_grid, dt=2e-4, E=200.0, nu=0.2, ptype=0, material_model=0, gravity=(0.0, -1.0, 0.0),
                        ground_friction=0.0, collision_type=2, n_controllers=0, max_steps=max_steps, precision=precision, device=device,
                        yield_stress=2.4)
    sheet = dict(friction=1.0, softness=666.0, cloth_force_scale=1.0, mpm_force_scale=1.0, sticky=True)
    rigid = None if palm is None else dict(table_spec(palm), friction=0.6, softness=666.0, contact=True)
    top = 0.4 + 0.1 * dx + h
    s13 = np.concatenate([[0.5, top + 0.15 - 0.002, 0.5], [1.0, 0.0, 0.0, 0.0], [0.0, -0.2, 0.0], [0.0, 0.0, 0.0]])   # (the palm's half height is 0.15)
    mat_id = (x[:, 0] > 0.5).astype(np.int32)
    return cfg, 2e-3, np.hstack([x, v, F, C]), V, Fc, sheet, rigid, s13, mat_id, dict(E=100.0, nu=0.3, yield_stress=1.0)
