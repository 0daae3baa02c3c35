"""Slab decomposition of the MPM substep over the GPUs of one node.

No reference counterpart: the reference is single-device.  One process per GPU; rank r owns the particles
whose stencil base lies in its x-range and the grid planes they touch.  Particles only interact through the
grid, and the quadratic stencil reaches two planes past the owned range, so neighbouring slabs share
`nplanes = 2 (+ 2 per cell of tolerated drift)` x-planes.  Per substep each rank exchanges with its two
neighbours only (point-to-point: on an MI355X node every GPU pair has its own xGMI link, so this is
`ncclSend/ncclRecv` on one link per neighbour - never a ring all-reduce of the grid):

  forward   after P2G           : SUM of {m, p} partials on the shared planes               ("grid_in")
            after contact       : SUM of the contact corrections of v_out (v_out - v_mixed) ("grid_out", minus_mixed)
  backward  after g2p.grad      : SUM of grid_v_out.grad partials                           ("grid_out.grad")
            after contact.grad  : SUM of grid_v_mixed.grad partials                         ("grid_mixed.grad")

After each sum both ranks hold the complete values on the shared planes and run the per-node kernels
(grid_op, its adjoint) redundantly there, so no second trip is needed.  The backward pass restores the
forward grid from the per-frame checkpoint, hence needs no forward exchange.  ext_f and primitive-state
adjoints are per-rank partial sums; `allreduce_primitives` adds them once per env step, where the reference
consumes them (rigid_simulator.py:92-93, 203-208).

The engine behind `SlabRunner` is anything with `phase / grad_phase / halo_pack / halo_unpack_add /
new_buffer`: `HipSlabEngine` (the product: libsoftmac_hip through the C ABI) or, in the CPU tests, a
stand-in built on the oracle - the exchange logic under test is the same code.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist


class HipSlabEngine:
    """Adapter from MPMSimulator (C ABI) to the SlabRunner engine interface."""

    def __init__(self, sim, use_torch_stream=True):
        self.sim = sim
        self.n = sim.n_grid
        self.dtype = torch.float32 if sim.precision == 32 else torch.float64
        self.device = torch.device("cuda", sim.device)
        if use_torch_stream:
            # run the kernels on torch's current stream: RCCL ops issued through torch.distributed then order
            # with them without host synchronisation
            s = torch.cuda.current_stream(self.device).cuda_stream
            sim._h.call("smac_set_stream", C.c_void_p(s))

    def new_buffer(self, nplanes):
        return torch.empty((nplanes, self.n, self.n, 4), dtype=self.dtype, device=self.device)

    def phase(self, f, k):
        self.sim._push_contact_flags()
        self.sim._h.call("smac_substep_phase", int(f), int(k))

    def grad_phase(self, f, k, ext_f_grad=None):
        import numpy as np
        from . import _ffi
        e = None
        if ext_f_grad is not None and k == 0:
            e = _ffi.as_f64(np.stack([np.asarray(g, dtype=np.float64).reshape(6) for g in ext_f_grad]))
        self.sim._h.call("smac_substep_grad_phase", int(f), _ffi.dptr(e), int(k))

    def halo_pack(self, field, plane0, nplanes, buf, minus_mixed=0):
        self.sim._h.call("smac_halo_pack", field.encode(), int(plane0), int(nplanes), C.c_void_p(buf.data_ptr()), int(minus_mixed))

    def halo_unpack_add(self, field, plane0, nplanes, buf):
        self.sim._h.call("smac_halo_unpack_add", field.encode(), int(plane0), int(nplanes), C.c_void_p(buf.data_ptr()))


class SlabRunner:
    """Drives one rank's engine and its halo exchanges.

    left_plane0 / right_plane0: first shared plane (in THIS rank's grid indexing) with the left / right
    neighbour; the neighbour indexes the same physical planes from its own right_plane0 / left_plane0.
    """

    def __init__(self, engine, rank, world, left_plane0, right_plane0, nplanes=2, has_contact=True, group=None):
        self.e, self.rank, self.world = engine, rank, world
        self.left0, self.right0, self.np = int(left_plane0), int(right_plane0), int(nplanes)
        self.has_contact = bool(has_contact)
        self.group = group
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        self._buf = {}
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.stage_on_host = backend == "gloo"       # gloo moves host memory: stage device buffers through the CPU

    def _buffers(self, side):
        if side not in self._buf:
            self._buf[side] = (self.e.new_buffer(self.np), self.e.new_buffer(self.np))
        return self._buf[side]

    def exchange(self, field, minus_mixed=0):
        """SUM the partials of `field` on the shared planes with both neighbours."""
        sides = []
        if self.left is not None:
            sides.append(("L", self.left, self.left0))
        if self.right is not None:
            sides.append(("R", self.right, self.right0))
        if not sides:
            return
        ops, staged = [], []
        for side, peer, plane0 in sides:                       # pack every side BEFORE any unpack: partials, not totals
            send, recv = self._buffers(side)
            self.e.halo_pack(field, plane0, self.np, send, minus_mixed)
            if self.stage_on_host and send.is_cuda:
                hs, hr = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
                staged.append((recv, hr))
                send, recv = hs, hr
            ops.append(dist.P2POp(dist.isend, send, peer, self.group))
            ops.append(dist.P2POp(dist.irecv, recv, peer, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for dev, host in staged:
            dev.copy_(host)
        for side, peer, plane0 in sides:
            self.e.halo_unpack_add(field, plane0, self.np, self._buffers(side)[1])

    def substep(self, f):
        self.e.phase(f, 0)
        self.exchange("grid_in")
        self.e.phase(f, 1)
        if self.has_contact:
            self.exchange("grid_out", minus_mixed=1)
        self.e.phase(f, 2)

    def substep_grad(self, f, ext_f_grad=None):
        self.e.grad_phase(f, 0, ext_f_grad)
        self.exchange("grid_out.grad")
        self.e.grad_phase(f, 1)
        if self.has_contact:
            self.exchange("grid_mixed.grad")
        self.e.grad_phase(f, 2)

    def run_substeps(self, f0, count):
        for f in range(f0, f0 + count):
            self.substep(f)

    def run_substeps_grad(self, f0, count, ext_f_grad=None):
        for f in range(f0 + count - 1, f0 - 1, -1):
            self.substep_grad(f, ext_f_grad)


class _DevArray:
    """A library-owned device buffer as seen by torch (zero copy, __cuda_array_interface__)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class PrimitiveReducer:
    """SUM over the slabs of the per-rank partial wrench sums `ext_f` and primitive-state adjoints, in place in the library's
    device buffers - once per env step, where the reference consumes them (rigid_simulator.py:92-93, 203-208).  Small
    all-reduces (6 P and 13 P substeps scalars), not grid traffic."""

    def __init__(self, sim, group=None):
        self.sim, self.group = sim, group
        self.P = len(sim.primitives)
        self.frames = sim.max_steps
        dev = torch.device("cuda", sim.device)
        self.gloo = dist.get_backend(group) == "gloo"

        def view(field):
            p, n, b = C.c_void_p(), C.c_int64(0), C.c_int32(0)
            sim._h.call("smac_grid_device_ptr", field.encode(), C.byref(p), C.byref(n), C.byref(b))
            assert b.value == 8
            return torch.as_tensor(_DevArray(p.value, n.value, "<f8"), device=dev)

        self.ext_f = view("ext_f") if self.P else None
        self.pgrad = view("prim_state.grad").view(max(self.P, 1), self.frames, 13) if self.P else None

    def _sum(self, t):
        if self.gloo:                                        # gloo moves host memory (CPU tests / ranks sharing one GPU)
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def allreduce_ext_f(self, clear=False):
        """every rank ends with the total wrench; `clear` then zeroes it as the reference's clear_ext_f does after the read"""
        if self.ext_f is None:
            return None
        self.sim.sync()
        self._sum(self.ext_f)
        total = self.ext_f.clone()
        if clear:
            self.ext_f.zero_()
        return total

    def allreduce_state_grad(self, f0, f1):
        if self.pgrad is None:
            return
        self.sim.sync()
        sl = self.pgrad[:, f0:f1].contiguous()
        self._sum(sl)
        self.pgrad[:, f0:f1] = sl


def allreduce_primitives(tensors, group=None):
    """Sum per-rank partials of ext_f / primitive-state adjoints (small, once per env step)."""
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return tensors
