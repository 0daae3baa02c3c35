"""Build libsoftmac_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m softmac_amd.build [--force] [--report]

The shared library is written in-tree to softmac_amd/lib/ (git-ignored, but it travels with
gpurun snapshots).  `--report` also stores hipcc's per-kernel resource usage
(VGPRs / scratch / occupancy) in softmac_amd/lib/resource_usage.txt.
"""
from __future__ import annotations

import os
import pathlib
import re
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
LIBDIR = ROOT / "lib"
LIB = LIBDIR / "libsoftmac_hip.so"
SOURCES = [CSRC / "softmac_hip.hip"]
DEPS = sorted(CSRC.glob("*.hpp")) + [ROOT.parent / "include" / "softmac_hip.h"]
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def is_stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + DEPS)


def build(force: bool = False, report: bool = False, verbose: bool = True, defines=(), out=None, extra=()) -> pathlib.Path:
    """`defines` / `out`: experiment builds of the same ABI (e.g. defines=["SMAC_CONST_F64=1"], out="libsoftmac_hip_cf64.so"),
    loaded with SMAC_LIB=<path> (tools/ab.sh, tools/prec_probe.py)."""
    if out is not None:
        return _compile(LIBDIR / out, list(defines), report, verbose, list(extra))
    if not force and not report and not is_stale():
        return LIB
    return _compile(LIB, list(defines), report, verbose, list(extra))


def _compile(LIB, defines, report, verbose, extra=()):
    LIBDIR.mkdir(exist_ok=True)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-ffp-contract=fast",
           "-fno-slp-vectorize",      # the SLP packer's v_pk_* f32 ops cost more v_mov shuffles than they save (measured +12%)
           "-ffast-math",             # the reference runs Taichi with fast_math=True (taichi_env.py:13); +4.5 %, all parity tolerances unchanged
           "-fno-finite-math-only",   # ... but NaN / Inf keep their meaning: an exploded state must stay visibly non-finite (tile_scale)
           "-Wno-unused-value", "-shared", "-fPIC", "-o", str(LIB)] + [f"-D{d}" for d in defines] + list(extra) + [str(s) for s in SOURCES]
    if report:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    if verbose:
        print("[softmac_amd.build]", " ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building libsoftmac_hip.so")
    if report:
        rows, cur = [], None
        for line in res.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
            for key in ("VGPRs", "AGPRs", "ScratchSize \\[bytes/lane\\]", "Occupancy \\[waves/SIMD\\]", "LDS Size \\[bytes/block\\]", "SGPRs"):
                m = re.search(rf"remark:\s+{key}: (\d+)", line)
                if m and cur is not None:
                    cur[key.replace("\\", "")] = m.group(1)
        out = LIBDIR / "resource_usage.txt"
        with open(out, "w") as fh:
            for r in rows:
                name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip() or r["name"]
                if "smac::" not in name:
                    continue                      # (rocPRIM's scan / transform trampolines for every arch it knows)
                fh.write(f"{name}: " + ", ".join(f"{k}={v}" for k, v in r.items() if k != "name") + "\n")
        if verbose:
            print(out.read_text())
    return LIB


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")]
    extra = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--flag=")]      # e.g. --flag=-fno-fast-math (later flags win)
    build(force="--force" in sys.argv, report="--report" in sys.argv, defines=defs, out=outs[0] if outs else None, extra=extra)
