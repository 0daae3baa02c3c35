"""ctypes wrapper of oracle/mpm_cpu.cpp (the plain C++/OpenMP f64 oracle port).  TEST INFRASTRUCTURE ONLY:
used by tests/test_cpu_port.py and by bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import pathlib
import subprocess

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
dp = C.POINTER(C.c_double)


class McParams(C.Structure):
    _fields_ = [("N", C.c_int), ("n_grid", C.c_int), ("substeps", C.c_int), ("ptype", C.c_int), ("model", C.c_int), ("P", C.c_int),
                ("sticky", C.c_int), ("collision_type", C.c_int), ("dt", C.c_double), ("mu", C.c_double), ("lam", C.c_double),
                ("p_vol", C.c_double), ("p_mass", C.c_double), ("g", C.c_double * 3)]


class McPrim(C.Structure):
    _fields_ = [("sdf", dp), ("normal", dp), ("res", C.c_int * 3), ("contact", C.c_int), ("lower", C.c_double * 3),
                ("upper", C.c_double * 3), ("sdf_dx", C.c_double), ("friction", C.c_double), ("softness", C.c_double)]


def _cpu_stamp():
    """The library is built with -march=native: a copy that travelled from another machine must be rebuilt."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                import hashlib
                return hashlib.sha1(line.encode()).hexdigest()
    except OSError:
        pass
    return "unknown"


def cpu_share():
    """CPUs this process may really use: the cgroup quota when there is one (a GPU box shows 256 CPUs and grants 16), else the affinity mask"""
    import os
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per))))
        except (OSError, ValueError):
            pass
    return n


def load():
    so = HERE / "_build" / "libmpm_cpu.so"
    stamp = HERE / "_build" / "cpu.stamp"
    stale = not so.exists() or so.stat().st_mtime < (HERE / "mpm_cpu.cpp").stat().st_mtime
    if not stale and (not stamp.exists() or stamp.read_text() != _cpu_stamp()):
        stale = True
    if stale:
        subprocess.check_call(["make", "-C", str(HERE), "-s", "-B"])
        stamp.write_text(_cpu_stamp())
    lib = C.CDLL(str(so))
    lib.mc_threads.restype = C.c_int
    return lib


def P(a):
    return None if a is None else a.ctypes.data_as(dp)


class CpuPort:
    """One-substep forward / adjoint on AOS float64 numpy arrays."""

    def __init__(self, sp, specs=()):
        """sp: oracle.softmac_oracle.SimParams; specs: list of dict(sdf, normal, lower, upper, dx, friction, softness, contact)."""
        self.lib = load()
        self.sp = sp
        self.m = McParams(0, sp.n_grid, sp.substeps, sp.ptype, sp.material_model, len(specs), 1 if sp.ground_friction >= 10 else 0,
                          sp.collision_type, sp.dt, sp.mu, sp.lam, sp.p_vol, sp.p_mass, (C.c_double * 3)(*sp.gravity))
        self._keep = []
        arr = (McPrim * max(len(specs), 1))()
        for i, s in enumerate(specs):
            sdf = np.ascontiguousarray(s["sdf"], dtype=np.float64); nrm = np.ascontiguousarray(s["normal"], dtype=np.float64)
            self._keep += [sdf, nrm]
            arr[i].sdf, arr[i].normal = P(sdf), P(nrm)
            arr[i].res = (C.c_int * 3)(*[int(r) for r in sdf.shape])
            arr[i].contact = 1 if s.get("contact", True) else 0
            arr[i].lower = (C.c_double * 3)(*np.asarray(s["lower"], dtype=float)); arr[i].upper = (C.c_double * 3)(*np.asarray(s["upper"], dtype=float))
            arr[i].sdf_dx, arr[i].friction, arr[i].softness = float(s["dx"]), float(s.get("friction", 0.9)), float(s.get("softness", 666.0))
        self.prims = arr
        self.nP = len(specs)

    def threads(self):
        return self.lib.mc_threads()

    def set_threads(self, n):
        self.lib.mc_set_threads(int(n))

    def substep(self, f, x, v, Cm, F, pst=None):
        N = len(x)
        self.m.N = N
        x, v, Cm, F = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, v, Cm, F))
        nx, nv, nC, nF = np.zeros_like(x), np.zeros_like(v), np.zeros_like(Cm), np.zeros_like(F)
        ext = np.zeros((max(self.nP, 1), 6))
        pst = np.zeros((max(self.nP, 1), 13)) if pst is None else np.ascontiguousarray(pst, dtype=np.float64)
        self.lib.mc_substep(C.byref(self.m), self.prims, P(pst), int(f), P(x), P(v), P(Cm), P(F), P(nx), P(nv), P(nC), P(nF), P(ext))
        return nx, nv, nC, nF, ext[:self.nP]

    def substep_grad(self, f, x, v, Cm, F, gx1, gv1, gC1, gF1, pst=None, ext_f_grad=None):
        N = len(x)
        self.m.N = N
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, v, Cm, F, gx1, gv1, gC1, gF1)]
        gx0, gv0, gC0, gF0 = np.zeros((N, 3)), np.zeros((N, 3)), np.zeros((N, 3, 3)), np.zeros((N, 3, 3))
        gp = np.zeros((max(self.nP, 1), 13))
        pst = np.zeros((max(self.nP, 1), 13)) if pst is None else np.ascontiguousarray(pst, dtype=np.float64)
        eg = None if ext_f_grad is None else np.ascontiguousarray(np.stack(ext_f_grad), dtype=np.float64)
        self.lib.mc_substep_grad(C.byref(self.m), self.prims, P(pst), int(f), *[P(a) for a in arrs], P(eg), P(gx0), P(gv0), P(gC0), P(gF0), P(gp))
        return gx0, gv0, gC0, gF0, gp[:self.nP]
