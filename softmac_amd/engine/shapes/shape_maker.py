"""Shapes - seeded particle clouds from the SHAPES config list
(/root/reference/softmac/engine/shapes/shape_maker.py:11-85): box / sphere sampling with numpy seed 0,
predefined (N,3) or (N,24) state arrays.  Pure numpy, host side."""
import numpy as np

COLORS = [(127 << 16) + 127, (127 << 8), 127, 127 << 16]


class Shapes:
    def __init__(self, cfg):
        self.objects, self.colors, self.dim = [], [], 3
        state = np.random.get_state()
        np.random.seed(0)                                        # reference :19-20
        for i in cfg:
            kwargs = {k: (eval(v) if isinstance(v, str) else v) for k, v in i.items() if k != 'shape'}
            if i['shape'] == 'box':
                self.add_box(**kwargs)
            elif i['shape'] == 'sphere':
                self.add_sphere(**kwargs)
            elif i['shape'] == 'predefined':
                self.add_predefined(**kwargs)
            else:
                raise NotImplementedError(f"Shape {i['shape']} is not supported!")
        np.random.set_state(state)

    def get_n_particles(self, volume):
        return max(int(volume / 0.2 ** 3) * 10000, 1)

    def add_object(self, particles, color=None, init_rot=None):
        if init_rot is not None:
            w, x, y, z = init_rot                                # quaternion -> rotation matrix (transforms3d.quat2mat)
            q = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            origin = particles.mean(axis=0)
            particles[:, :self.dim] = (particles[:, :self.dim] - origin[:self.dim]) @ q.T + origin[:self.dim]
        self.objects.append(particles)
        if color is None or isinstance(color, int):
            tmp = COLORS[(len(self.objects) - 1) % len(COLORS)] if color is None else color
            color = np.full(len(particles), tmp, np.int32)
        self.colors.append(color)

    def add_box(self, init_pos, width, n_particles=10000, color=None, init_rot=None):
        width = np.array([width] * self.dim) if isinstance(width, float) else np.array(width)
        if n_particles is None:
            n_particles = self.get_n_particles(np.prod(width))
        p = (np.random.random((n_particles, self.dim)) * 2 - 1) * (0.5 * width) + np.array(init_pos)
        self.add_object(p, color, init_rot=init_rot)

    def add_sphere(self, init_pos, radius, n_particles=10000, color=None, init_rot=None):
        if n_particles is None:
            n_particles = self.get_n_particles((radius ** 3) * 4 * np.pi / 3)
        p = np.random.normal(size=(n_particles, self.dim))
        p /= np.linalg.norm(p, axis=-1, keepdims=True)
        u = np.random.random(size=(n_particles, 1)) ** (1. / self.dim)
        self.add_object(p * u * radius + np.array(init_pos)[:self.dim], color, init_rot=init_rot)

    def add_predefined(self, path=None, offset=None, color=None, state=None):
        p = np.array(state, dtype=np.float64) if state is not None else np.load(path)
        p[:, :self.dim] += np.zeros(self.dim) if offset is None else np.asarray(offset)
        self.add_object(p, color)

    def get(self):
        assert len(self.objects) > 0, "please add at least one shape into the scene"
        return np.concatenate(self.objects), np.concatenate(self.colors)
