#!/usr/bin/env python3
"""Generate committed golden vectors from the f64 oracle (run here, in the build container).

Each tests/golden/oracle_<scene>.npz holds the inputs of a short rollout (reference data files as
initial state where one exists) and the oracle's outputs: the last frame, the accumulated ext_f, the
adjoints at frame 0 for fixed random seeds, and the primitive-state adjoints.  tests/test_oracle.py
re-derives them on the CPU (oracle regression pin); tests/test_gpu_parity.py compares the HIP path
against them on the GPU box, where neither /root/reference nor any generator is needed.
"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import helpers as H  # noqa: E402
import scenes_golden as G  # noqa: E402


def main():
    for name in G.SCENES:
        sc = G.build(name)
        out = G.run_oracle(sc)
        path = H.GOLDEN / f"oracle_{name}.npz"
        np.savez_compressed(path, **out)
        print(name, path.stat().st_size, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
