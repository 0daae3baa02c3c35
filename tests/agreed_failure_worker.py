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


class Runner:                                              # stands for LibSlabRunner: what matters here is that abort() is called on every rank
    aborted = False

    def abort(self):
        self.aborted = True


run = Runner()
err = None
try:
    if mode == "rank1_fails" and rank == 1:
        raise RuntimeError("smac_substeps_slab failed (-2): a particle left the halo of its grid block [rank 1: the RCCL communicator was aborted]")
except Exception as e:                                     # noqa: BLE001
    err = f"{type(e).__name__}: {e}"
raised = None
try:
    parallel.agreed_failure(err, run)
except RuntimeError as e:
    raised = str(e)
json.dump({"raised": raised, "aborted": run.aborted}, open(os.path.join(out, f"rank{rank}.json"), "w"))
dist.barrier()
dist.destroy_process_group()
