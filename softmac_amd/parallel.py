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

import numpy as np
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

    # ---- migration (rows = x3 v3 F9 C9, the layout of get_state)
    def set_segment(self, n_live, shift):
        self.sim.set_segment(n_live, shift)

    def get_state(self, f):
        return self.sim.get_state(f)

    def set_state(self, f, st):
        n = len(st)
        self.sim.set_state(f, (st[:, 0:3], st[:, 3:6], st[:, 6:15].reshape(n, 3, 3), st[:, 15:24].reshape(n, 3, 3)))

    def get_grad_rows(self, f):
        gx, gv, gF, gC = self.sim.get_grad_full(f)
        n = len(gx)
        return np.hstack([gx, gv, gF.reshape(n, 9), gC.reshape(n, 9)])

    def add_grad_rows(self, f, g):
        n = len(g)
        self.sim.add_grad(f, gx=g[:, 0:3], gv=g[:, 3:6], gF=g[:, 6:15].reshape(n, 3, 3), gC=g[:, 15:24].reshape(n, 3, 3))


class SlabRunner:
    """Drives one rank's engine and its halo exchanges.

    left_plane0 / right_plane0: first shared plane (in THIS rank's grid indexing) with the left / right
    neighbour; the neighbour indexes the same physical planes from its own right_plane0 / left_plane0.

    Particle MIGRATION: `own=(lo, hi)` is the range of stencil bases this rank owns and `ids` the global ids of its
    particles.  `migrate(f)` - called by the driver at a re-sort / env-step boundary, never inside a substep - hands the
    particles whose base left [lo, hi) to the neighbour (neighbour-only variable-size send/recv) and starts a new SEGMENT at
    handle frame f+1 with the new particle set; the tape keeps both orderings of the migration point (frames f and f+1), and
    `migrate_grad()` carries the adjoint of frame f+1 back across the slab boundary in the backward sweep."""

    def __init__(self, engine, rank, world, left_plane0, right_plane0, nplanes=2, has_contact=True, group=None, own=None, ids=None):
        self.e, self.rank, self.world = engine, rank, world
        self.left0, self.right0, self.np = int(left_plane0), int(right_plane0), int(nplanes)
        # has_contact: bool, or (left, right) - whether a contact primitive can touch the planes shared with that neighbour (`contact_sides`);
        # a side without contact skips the two contact exchanges (v_out corrections, grid_v_mixed.grad) with that neighbour
        if isinstance(has_contact, (tuple, list)):
            self.contact_side = {"L": bool(has_contact[0]), "R": bool(has_contact[1])}
        else:
            self.contact_side = {"L": bool(has_contact), "R": bool(has_contact)}
        self.has_contact = self.contact_side["L"] or self.contact_side["R"]
        self.group = group
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        self._buf = {}
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.stage_on_host = backend == "gloo"       # gloo moves host memory: stage device buffers through the CPU
        self.own = None if own is None else (int(own[0]), int(own[1]))
        self.ids = None if ids is None else np.asarray(ids, dtype=np.int64).copy()
        self.shift = 0                               # duplicate (migration) frames before the current segment
        self.migrations = []

    def _buffers(self, side):
        if side not in self._buf:
            self._buf[side] = (self.e.new_buffer(self.np), self.e.new_buffer(self.np))
        return self._buf[side]

    def exchange(self, field, minus_mixed=0, contact_only=False):
        """SUM the partials of `field` on the shared planes with both neighbours (contact_only: only where a primitive reaches those planes)."""
        sides = []
        if self.left is not None and (self.contact_side["L"] or not contact_only):
            sides.append(("L", self.left, self.left0))
        if self.right is not None and (self.contact_side["R"] or not contact_only):
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
            self.exchange("grid_out", minus_mixed=1, contact_only=True)
        self.e.phase(f, 2)

    def substep_grad(self, f, ext_f_grad=None):
        self.e.grad_phase(f, 0, ext_f_grad)
        self.exchange("grid_out.grad")
        self.e.grad_phase(f, 1)
        if self.has_contact:
            self.exchange("grid_mixed.grad", contact_only=True)
        self.e.grad_phase(f, 2)

    def run_substeps(self, f0, count):
        for f in range(f0, f0 + count):
            self.substep(f)

    def run_substeps_grad(self, f0, count, ext_f_grad=None):
        for f in range(f0 + count - 1, f0 - 1, -1):
            self.substep_grad(f, ext_f_grad)

    # ---- migration ------------------------------------------------------------------------------------------
    def _exchange_rows(self, to_left, to_right):
        """variable-size neighbour exchange of (k, c) float64 rows: returns (rows from the left, rows from the right)"""
        cols = to_left.shape[1]
        dev = "cpu" if (self.stage_on_host or not dist.is_initialized()) else "cuda"
        peers = [(self.left, to_left), (self.right, to_right)]
        cnt_out = [torch.tensor([len(r)], dtype=torch.int64, device=dev) for _, r in peers]
        cnt_in = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in peers]
        ops = []
        for (peer, _), co, ci in zip(peers, cnt_out, cnt_in):
            if peer is not None:
                ops += [dist.P2POp(dist.isend, co, peer, self.group), dist.P2POp(dist.irecv, ci, peer, self.group)]
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
        bufs_in = [torch.empty((int(ci.item()), cols), dtype=torch.float64, device=dev) for ci in cnt_in]
        ops, keep = [], []
        for (peer, rows), bi in zip(peers, bufs_in):
            if peer is None:
                continue
            if len(rows):
                t = torch.as_tensor(np.ascontiguousarray(rows), dtype=torch.float64).to(dev)
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, peer, self.group))
            if len(bi):
                ops.append(dist.P2POp(dist.irecv, bi, peer, self.group))
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
        return bufs_in[0].cpu().numpy(), bufs_in[1].cpu().numpy()

    def migrate(self, f):
        """Hand over the particles of frame f whose stencil base (mpm_simulator.py:215) left this rank's range.  The next
        segment starts at handle frame f + 1 (returned) and holds the kept particles followed by the arrivals."""
        assert self.own is not None and self.ids is not None, "SlabRunner(own=..., ids=...) is needed for migration"
        st = self.e.get_state(f)
        base = np.floor(st[:, 0] * self.e.n - 0.5).astype(np.int64)
        to_l = (base < self.own[0]) if self.left is not None else np.zeros(len(st), dtype=bool)
        to_r = (base >= self.own[1]) if self.right is not None else np.zeros(len(st), dtype=bool)
        keep = ~(to_l | to_r)
        pack = lambda m: np.hstack([st[m], self.ids[m, None].astype(np.float64)])
        from_l, from_r = self._exchange_rows(pack(to_l), pack(to_r))
        new = np.vstack([st[keep], from_l[:, :24], from_r[:, :24]])
        new_ids = np.concatenate([self.ids[keep], from_l[:, 24].astype(np.int64), from_r[:, 24].astype(np.int64)])
        assert len(new) >= 1, "a slab lost all its particles"
        self.migrations.append(dict(frame=f, n_old=len(st), keep=np.nonzero(keep)[0], sl=np.nonzero(to_l)[0], sr=np.nonzero(to_r)[0],
                                    nl=len(from_l), nr=len(from_r), ids_old=self.ids))
        self.shift += 1
        self.e.set_segment(len(new), self.shift)
        self.e.set_state(f + 1, new)
        self.ids = new_ids
        return f + 1

    def migrate_grad(self):
        """Backward of the most recent `migrate`: the adjoint of the later segment's first frame goes back to the frame it was
        copied from - across the slab boundary for the particles that crossed it."""
        rec = self.migrations.pop()
        f = rec["frame"]
        g = self.e.get_grad_rows(f + 1)
        nk = len(rec["keep"])
        back_l, back_r = self._exchange_rows(g[nk:nk + rec["nl"]], g[nk + rec["nl"]:])
        old = np.zeros((rec["n_old"], 24))
        old[rec["keep"]] = g[:nk]
        old[rec["sl"]] = back_l
        old[rec["sr"]] = back_r
        self.shift -= 1
        self.e.set_segment(rec["n_old"], self.shift)
        self.e.add_grad_rows(f, old)
        self.ids = rec["ids_old"]
        return f


def comm_unique_id():
    """128-byte RCCL id made by the library on the calling rank (ncclGetUniqueId through the C ABI)."""
    from . import _ffi
    lib = _ffi.load_library()
    buf = C.create_string_buffer(128)
    rc = lib.smac_comm_unique_id(buf)
    if rc != 0:
        msg = lib.smac_last_error(None)
        raise _ffi.SmacError(f"smac_comm_unique_id failed ({rc}): {msg.decode() if msg else ''}")
    return buf.raw


def rendezvous_unique_id(rank, group=None):
    """rank 0 makes the id, every rank of the (CPU, gloo) process group receives it.  A failure on rank 0 (RCCL not loadable) travels instead of the
    id and is raised on EVERY rank - a rank 0 that raised before the broadcast would leave the others waiting in it."""
    box = [None]
    if rank == 0:
        try:
            box = [comm_unique_id()]
        except Exception as e:                               # noqa: BLE001 - whatever it is, the other ranks must hear of it
            box = [RuntimeError(f"rank 0 could not create the RCCL id: {type(e).__name__}: {e}")]
    dist.broadcast_object_list(box, src=0, group=group)
    if isinstance(box[0], Exception):
        raise box[0]
    return box[0]


def all_ranks_ok(error, group=None):
    """error: None or this rank's failure (str).  Returns the list of failures of all ranks (empty = every rank is fine): the ranks decide TOGETHER
    whether a collective resource came up, so that none of them walks on into an exchange its neighbour never posts."""
    world = dist.get_world_size(group)
    allf = [None] * world
    dist.all_gather_object(allf, error, group=group)
    return [f"rank {r}: {e}" for r, e in enumerate(allf) if e]


class FailureWatch:
    """Out-of-band failure channel between the ranks of ONE node (the metric's 8 GPUs are one node), without a collective.

    Why: a rank that fails inside the slab loop aborts ITS communicator (softmac_hip.hip slab_guard), but ncclCommAbort there does not release
    the receives its neighbours have enqueued: they sit in their next stream synchronisation and never reach a collective in which they could
    be told (`agreed_failure` alone deadlocks until gloo's timeout).  So the failing rank PUBLISHES - one small file in a directory
    every rank derives from the launcher's rendezvous (MASTER_ADDR / MASTER_PORT), written atomically - and every rank runs a daemon thread that
    polls the directory; when another rank's file appears the thread calls `runner.abort()` (smac_comm_abort, safe from a second thread: the
    library takes its communicator mutex) so that the main thread's pending wait returns, and remembers the message.  `arm(seconds, what)` adds a
    deadline for one bounded step (bench.py: the first warm-up window, the first place an N-rank run can hang): when it passes, the thread
    publishes, aborts and ends the process with status 3 - a fresh, non-zero exit instead of a hang; nothing is re-executed.

    Supported reactions, in order of preference: (1) the main thread's call raises (its exchange finds no communicator), the caller passes the
    error to `agreed_failure(error, runner, watch=watch)` and every rank raises the same RuntimeError; (2) under a launcher, exit non-zero."""

    def __init__(self, rank, world, runner=None, directory=None, key=None, poll=0.05):
        import os
        import tempfile
        import threading
        self.rank, self.world, self.runner, self.poll = int(rank), int(world), runner, float(poll)
        if key is None:
            key = f"{os.environ.get('MASTER_ADDR', 'local')}-{os.environ.get('MASTER_PORT', '0')}-{os.environ.get('TORCHELASTIC_RUN_ID', 'run')}"
            # a token of THIS run, made by rank 0 and handed round while every rank is still healthy: a file a crashed earlier run left under the same
            # rendezvous address must not look like a failure of this one
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() == self.world:
                import uuid
                box = [uuid.uuid4().hex if self.rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                key += "-" + str(box[0])
        base = directory or os.environ.get("SMAC_FAIL_DIR") or tempfile.gettempdir()
        self.dir = os.path.join(base, "smac-fail-" + "".join(c if c.isalnum() or c in "-_." else "_" for c in str(key)))
        os.makedirs(self.dir, exist_ok=True)
        self._mine = os.path.join(self.dir, f"rank{self.rank}.txt")
        try:
            os.unlink(self._mine)                            # a stale file of an earlier run with the same rendezvous
        except OSError:
            pass
        self.failure = None                                  # "rank r: message" of the first OTHER rank seen to have failed
        self._deadline = None
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._loop, name="smac-failure-watch", daemon=True)
        self._thread.start()

    def publish(self, message):
        """this rank failed: tell the others (atomic rename, so a reader never sees half a message)"""
        import os
        tmp = self._mine + ".tmp"
        with open(tmp, "w") as fh:
            fh.write(str(message)[:2000])
        os.replace(tmp, self._mine)

    def arm(self, seconds, what="a bounded step"):
        import time
        self._deadline = (time.monotonic() + float(seconds), str(what))

    def disarm(self):
        self._deadline = None

    def _others(self):
        import os
        out = []
        try:
            names = os.listdir(self.dir)
        except OSError:
            return out
        for n in sorted(names):
            if n.startswith("rank") and n.endswith(".txt") and n != os.path.basename(self._mine):
                try:
                    out.append(f"rank {int(n[4:-4])}: " + open(os.path.join(self.dir, n)).read())
                except (OSError, ValueError):
                    pass
        return out

    def _abort_runner(self):
        if self.runner is not None and hasattr(self.runner, "abort"):
            try:
                self.runner.abort()
            except Exception:                                # noqa: BLE001 - the failure being reported is the one that matters
                pass

    def _loop(self):
        import os
        import sys
        import time
        while not self._stop.wait(self.poll):
            others = self._others()
            if others:
                self.failure = "; ".join(others)
                self._abort_runner()                         # releases this rank's pending receive / barrier wait
                return
            dl = self._deadline
            if dl is not None and time.monotonic() > dl[0]:
                msg = f"{dl[1]} did not finish within its time limit on rank {self.rank}"
                self.publish(msg)
                self._abort_runner()
                print("softmac_amd.parallel.FailureWatch: " + msg + " - exiting with status 3", file=sys.stderr, flush=True)
                os._exit(3)

    def check(self):
        if self.failure:
            raise RuntimeError("collective run failed: " + self.failure)

    def close(self):
        import os
        self._stop.set()
        self._thread.join(timeout=2.0)
        try:
            os.unlink(self._mine)
        except OSError:
            pass
        try:
            os.rmdir(self.dir)                               # (the last rank to leave succeeds)
        except OSError:
            pass


def agreed_failure(error, runner=None, group=None, watch=None):
    """End of a window of collective work for hosts that keep their ranks alive: every rank reports None or its failure; if ANY rank failed,
    every rank aborts its communicator (`runner.abort()`) and raises the same RuntimeError naming the ranks that failed.

    This is a COLLECTIVE (all_gather over the control group), so every rank must be able to reach it.  A healthy rank whose neighbour failed in
    the middle of the slab loop cannot by itself: its next stream synchronisation waits on a receive nobody will answer (ncclCommAbort on the
    failing rank does not release it).  Pass a `FailureWatch`: the failing rank publishes out of band BEFORE entering the all_gather, the healthy
    ranks' watch threads abort their communicators, their pending calls return with an error, and they arrive here with it.  Without a watch the
    only supported reaction to a failure inside the loop is a non-zero exit under a launcher, which then stops the other ranks - that is what
    bench.py does (it does not call this function)."""
    if watch is not None:
        if error:
            watch.publish(error)
        elif watch.failure:
            error = f"stopped because another rank failed ({watch.failure})"
    failed = all_ranks_ok(error, group=group)
    if failed:
        if runner is not None and hasattr(runner, "abort"):
            try:
                runner.abort()
            except Exception:                                # noqa: BLE001 - the failure being reported is the one that matters
                pass
        raise RuntimeError("collective run failed: " + "; ".join(failed))


def agree_contact_sides(sides, rank, world, group=None):
    """`contact_sides` is evaluated per rank; a boundary's two ranks must take the same decision or one of them posts an exchange the other
    does not: every rank learns every rank's flags and a boundary uses the OR of its two sides."""
    allf = [None] * world
    dist.all_gather_object(allf, (bool(sides[0]), bool(sides[1])), group=group)
    left = rank > 0 and (allf[rank][0] or allf[rank - 1][1])
    right = rank < world - 1 and (allf[rank][1] or allf[rank + 1][0])
    return bool(left), bool(right)


class LibSlabRunner:
    """The slab loop inside the library (smac_substeps_slab[_grad]): RCCL send / recv between neighbours on the library's own communication
    stream, the phases and the plane pack / unpack enqueued from C++ - the per-substep Python of `SlabRunner` is gone (round 2 measured 62 us of
    host time per backward substep in it).  Same `run_substeps / run_substeps_grad` surface, plus the primitives' reductions."""

    def __init__(self, sim, rank, world, left_plane0, right_plane0, nplanes=2, has_contact=(True, True), own=None, unique_id=None, self_loop=False,
                 drift_tol=None):
        self.sim, self.rank, self.world = sim, rank, world
        if unique_id is None:
            unique_id = comm_unique_id()
        sim._h.call("smac_comm_init", unique_id, int(rank), int(world))
        cl, cr = (bool(has_contact), bool(has_contact)) if not isinstance(has_contact, (tuple, list)) else (bool(has_contact[0]), bool(has_contact[1]))
        tol = (nplanes - 2) // 2 if drift_tol is None else int(drift_tol)
        if own is None:
            lo, hi = 1, 0                                     # no range check
        else:
            lo = int(own[0]) - (tol if (rank > 0 or self_loop) else 10 ** 6)
            hi = int(own[1]) - 1 + (tol if (rank < world - 1 or self_loop) else 10 ** 6)
        sim._h.call("smac_comm_slab", int(left_plane0), int(right_plane0), int(nplanes), int(cl), int(cr), int(lo), int(hi), 1 if self_loop else 0)
        self._counts = []

    def run_substeps(self, f0, count):
        self.sim._push_contact_flags()
        self.sim._h.call("smac_substeps_slab", int(f0), int(count))

    def run_substeps_grad(self, f0, count, ext_f_grad=None):
        from . import _ffi
        self.sim._push_contact_flags()
        e = None
        if ext_f_grad is not None:
            e = _ffi.as_f64(np.stack([np.asarray(g, dtype=np.float64).reshape(6) for g in ext_f_grad]))
        self.sim._h.call("smac_substeps_slab_grad", int(f0), int(count), _ffi.dptr(e))

    # the primitives' reductions (PrimitiveReducer's surface)
    def allreduce_ext_f(self, clear=False):
        from . import _ffi
        P = len(self.sim.primitives)
        out = np.zeros((max(P, 1), 6))
        self.sim._h.call("smac_comm_allreduce_ext_f", _ffi.dptr(out), 1 if clear else 0)
        return out[:P]

    def allreduce_state_grad(self, f0, f1):
        self.sim._h.call("smac_comm_allreduce_prim_grad", int(f0), int(f1))

    def exchanges(self):
        return int(self.sim.get_param("exchanges"))

    # ---- migration on the device (smac_migrate / smac_migrate_grad); the tape of migrations lives in the library
    def set_ids(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        assert ids.shape == (self.sim.n_particles,)
        self.sim._h.call("smac_set_ids", ids.ctypes.data_as(C.POINTER(C.c_int64)))

    def ids(self):
        out = np.zeros(self.sim.n_particles, dtype=np.int64)
        self.sim._h.call("smac_get_ids", out.ctypes.data_as(C.POINTER(C.c_int64)))
        return out

    def migrate(self, f, own):
        """particles of frame f whose stencil base left own = [lo, hi) change hands; the next segment starts at frame f + 1 (returned)"""
        out = np.zeros(3, dtype=np.int32)
        self._counts.append(self.sim.n_particles)
        self.sim._h.call("smac_migrate", int(f), int(own[0]), int(own[1]), out.ctypes.data_as(C.POINTER(C.c_int32)))
        self.sim.n_particles = int(out[0])
        self.moved = getattr(self, "moved", 0) + int(out[1])
        return f + 1

    def migrate_grad(self):
        self.sim._h.call("smac_migrate_grad")
        self.sim.n_particles = self._counts.pop()

    def close(self):
        self.sim._h.call("smac_comm_destroy")

    def abort(self):
        """another rank failed inside the collective loop: drop the communicator without waiting for its pending receives (smac_comm_abort)"""
        self.sim._h.call("smac_comm_abort")


def contact_sides(specs, states, n_grid, left_plane0, right_plane0, nplanes, rank, world, band=5e-3, drift_cells=2.0):
    """(left, right): can a contact primitive put a correction on the x-planes shared with that neighbour?  Conservative: the primitive's
    SDF table box, rotated by its pose, at (a sample of) the frames of `states` ([frame][13] per primitive, or one 13-vector), grown
    by the contact band, the 2.5 cells a quadratic stencil reaches and the particle drift the binning tolerates.  Both neighbours of a boundary
    evaluate the same planes with the same primitive data, so they agree.  SMAC_SLAB_ALL_EXCHANGES=1 turns the shortcut off."""
    import os
    if os.environ.get("SMAC_SLAB_ALL_EXCHANGES"):
        return rank > 0, rank < world - 1
    dx = 1.0 / n_grid
    reach = band + (2.5 + drift_cells) * dx
    out = []
    for side, plane0, exists in (("L", left_plane0, rank > 0), ("R", right_plane0, rank < world - 1)):
        hit = False
        if exists:
            lo, hi = plane0 * dx, (plane0 + nplanes) * dx
            for spec, st in zip(specs, states):
                if not spec.get("contact", True):
                    continue
                lw, up = np.asarray(spec["lower"], dtype=np.float64), np.asarray(spec["upper"], dtype=np.float64)
                box = np.array([[a, b, c] for a in (lw[0], up[0]) for b in (lw[1], up[1]) for c in (lw[2], up[2])])
                st2 = np.atleast_2d(np.asarray(st, dtype=np.float64))
                for row in st2[:: max(1, len(st2) // 64)].tolist() + [st2[-1].tolist()]:
                    w, q = row[3], np.asarray(row[4:7])
                    n2 = w * w + q @ q
                    # x-extent of the rotated table box (first row of the rotation matrix of the pose quaternion)
                    r0 = np.array([w * w + q[0] * q[0] - q[1] * q[1] - q[2] * q[2], 2 * (q[0] * q[1] - w * q[2]), 2 * (q[0] * q[2] + w * q[1])]) / n2
                    ext = box @ r0
                    if row[0] + ext.max() + reach >= lo and row[0] + ext.min() - reach <= hi:
                        hit = True
                        break
        out.append(hit)
    return tuple(out)


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
