"""Particle clouds of a scene, from the SHAPES list of a config.

Behavioural contract (what a user of the reference's `Shapes` relies on, /root/reference/softmac/engine/shapes/shape_maker.py:11-85):
`Shapes(cfg.SHAPES).get()` returns `(particles, colors)`, particles `(N, 3)` - or `(N, 24)` state rows when a predefined file holds
them -, concatenated in the order of the config list; string-valued entries of a shape's dict are Python expressions (`"(0.5, 0.1, 0.5)"`);
and the sampled clouds are THE SAME numbers as the reference's for the same config.  The last point is data compatibility: the legacy numpy
generator is seeded with 0 once per scene (:19-20), a box draws one `random((n, 3))` block (:58), a ball draws `normal((n, 3))` then
`random((n, 1))` (:70-72), in list order, and the caller's generator state is put back afterwards (:18, 32).  Everything around that
sequence is this module's own: samplers are plain functions registered by shape name, a cloud is a small record, and `Shapes` only
assembles them.  Host side, pure numpy.
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass

import numpy as np

DIM = 3
PALETTE = ((127 << 16) + 127, 127 << 8, 127, 127 << 16)           # packed 0xRRGGBB defaults, one per object in turn
PARTICLES_PER_REFERENCE_CUBE = 10000                                # density used when a shape gives no particle count: 10^4 per 0.2^3


@dataclass
class Cloud:
    rows: np.ndarray                   # (n, 3) positions or (n, 24) state rows (x3 v3 F9 C9)
    color: object = None               # None -> palette, int -> one packed colour, array -> per particle
    init_rot: object = None            # quaternion (w, x, y, z) about the cloud's centroid, or None


@contextlib.contextmanager
def _scene_generator():
    """legacy global numpy generator, seeded 0 for the scene and handed back untouched"""
    saved = np.random.get_state()
    np.random.seed(0)
    try:
        yield
    finally:
        np.random.set_state(saved)


def _count_for(volume):
    return max(int(volume / 0.2 ** 3) * PARTICLES_PER_REFERENCE_CUBE, 1)


def _quat_matrix(q):
    w, x, y, z = (float(c) for c in q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def sample_box(init_pos, width, n_particles=10000, color=None, init_rot=None):
    half = 0.5 * (np.full(DIM, width) if isinstance(width, float) else np.asarray(width, dtype=np.float64))
    n = _count_for(np.prod(2 * half)) if n_particles is None else n_particles
    unit = np.random.random((n, DIM))                                # the one draw of a box
    return Cloud((unit * 2 - 1) * half + np.asarray(init_pos), color, init_rot)


def sample_sphere(init_pos, radius, n_particles=10000, color=None, init_rot=None):
    n = _count_for(radius ** 3 * 4 * np.pi / 3) if n_particles is None else n_particles
    direction = np.random.normal(size=(n, DIM))                     # first draw: directions
    direction /= np.linalg.norm(direction, axis=-1, keepdims=True)
    shell = np.random.random(size=(n, 1)) ** (1.0 / DIM)            # second draw: radii, uniform in volume
    return Cloud(direction * shell * radius + np.asarray(init_pos)[:DIM], color, init_rot)


def load_predefined(path=None, offset=None, color=None, state=None):
    rows = np.array(state, dtype=np.float64) if state is not None else np.load(path)
    if offset is not None:
        rows[:, :DIM] += np.asarray(offset)
    return Cloud(rows, color)


SAMPLERS = {"box": sample_box, "sphere": sample_sphere, "predefined": load_predefined}


class Shapes:
    def __init__(self, cfg):
        self.dim = DIM
        self.objects, self.colors = [], []
        with _scene_generator():
            for entry in cfg:
                kind = entry["shape"]
                if kind not in SAMPLERS:
                    raise NotImplementedError(f"Shape {kind} is not supported!")
                args = {k: (eval(v) if isinstance(v, str) else v) for k, v in entry.items() if k != "shape"}   # noqa: S307 - config files are code
                self._place(SAMPLERS[kind](**args))

    def _place(self, cloud):
        rows = cloud.rows
        if cloud.init_rot is not None:
            pivot = rows[:, :DIM].mean(axis=0)
            rows[:, :DIM] = (rows[:, :DIM] - pivot) @ _quat_matrix(cloud.init_rot).T + pivot
        color = cloud.color
        if color is None or isinstance(color, int):
            packed = PALETTE[len(self.objects) % len(PALETTE)] if color is None else color
            color = np.full(len(rows), packed, dtype=np.int32)
        self.objects.append(rows)
        self.colors.append(color)

    # the reference's incremental surface: a shape added after construction (draws from the caller's generator, as there)
    def add_box(self, *a, **k):
        self._place(sample_box(*a, **k))

    def add_sphere(self, *a, **k):
        self._place(sample_sphere(*a, **k))

    def add_predefined(self, *a, **k):
        self._place(load_predefined(*a, **k))

    def get(self):
        if not self.objects:
            raise AssertionError("please add at least one shape into the scene")
        return np.concatenate(self.objects), np.concatenate(self.colors)
