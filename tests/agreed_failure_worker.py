"""One rank of tests/test_slabs.py::test_run_time_failure_on_one_rank_surfaces_on_all: parallel.agreed_failure over gloo on CPU."""
import json
import os
import pathlib
import sys

import torch.distributed as dist

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
from softmac_amd import parallel  # noqa: E402

rank, world, port, out, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=rank, world_size=world)


import threading  # noqa: E402
import time  # noqa: E402


class Runner:                                              # stands for LibSlabRunner: what matters here is that abort() is called on every rank
    aborted = False

    def __init__(self):
        self.released = threading.Event()

    def abort(self):
        self.aborted = True
        self.released.set()

    def run_window(self):
        """a window whose exchange the neighbour never answers: the call sits in its stream synchronisation until the communicator is aborted
        (what smac_substeps_slab + a synchronisation do on a healthy rank whose neighbour failed); bounded here so that a broken test ends"""
        if not self.released.wait(timeout=60.0):
            raise TimeoutError("the pending exchange was never released")
        raise RuntimeError("smac_substeps_slab failed (-4): exchange: the communicator was aborted (smac_comm_abort)")


run = Runner()
watch = parallel.FailureWatch(rank, world, runner=run, directory=out) if mode == "rank1_fails_peer_pending" else None
err = None
t0 = time.monotonic()
try:
    if mode in ("rank1_fails", "rank1_fails_peer_pending") and rank == 1:
        raise RuntimeError("smac_substeps_slab failed (-2): a particle left the halo of its grid block [rank 1: the RCCL communicator was aborted]")
    if mode == "rank1_fails_peer_pending":
        run.run_window()                                   # rank 0: blocks until its watch thread has seen rank 1's file and aborted the runner
except Exception as e:                                     # noqa: BLE001
    err = f"{type(e).__name__}: {e}"
raised = None
try:
    parallel.agreed_failure(err, run, watch=watch)
except RuntimeError as e:
    raised = str(e)
json.dump({"raised": raised, "aborted": run.aborted, "seconds": time.monotonic() - t0}, open(os.path.join(out, f"rank{rank}.json"), "w"))
if watch is not None:
    watch.close()
dist.barrier()
dist.destroy_process_group()
