"""Face neighbourhoods for the penetration tracing (reference soft_cloth/engine/primitive/process_faces.py:5-53).

For every face: the first `n_neighbours` faces reached breadth-first over shared edges, each with a flag telling whether its
orientation is inverted relative to the face (the shared edge is traversed in the same direction by both).  Padded with the face
itself."""
from __future__ import annotations

import collections

import numpy as np


def process(faces, n_neighbours=100):
    faces = np.asarray(faces)
    n_faces = faces.shape[0]
    by_edge = collections.defaultdict(list)
    for i in range(n_faces):
        for j in range(3):
            a, b = int(faces[i, j]), int(faces[i, (j + 1) % 3])
            by_edge[(a, b) if a < b else (b, a)].append(i)
    directed = [set((int(faces[i, j]), int(faces[i, (j + 1) % 3])) for j in range(3)) for i in range(n_faces)]
    neighbours = np.zeros((n_faces, n_neighbours), dtype=np.int32)
    direction = np.zeros((n_faces, n_neighbours), dtype=np.int8)
    for i in range(n_faces):
        order = []
        seen = np.zeros(n_faces, dtype=bool)
        todo = collections.deque([(i, False)])
        while todo:
            cur, inverse = todo.popleft()
            if seen[cur]:
                continue
            order.append((cur, inverse))
            if len(order) > n_neighbours:
                break
            seen[cur] = True
            for j in range(3):
                a, b = int(faces[cur, j]), int(faces[cur, (j + 1) % 3])
                for other in by_edge[(a, b) if a < b else (b, a)]:
                    if other != cur:
                        todo.append((other, (not inverse) if (a, b) in directed[other] else inverse))
        order = order[1:] + [(i, False)] * max(0, n_neighbours - (len(order) - 1))
        neighbours[i] = [c for c, _ in order[:n_neighbours]]
        direction[i] = [1 if v else 0 for _, v in order[:n_neighbours]]
    return neighbours, direction
