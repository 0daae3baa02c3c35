"""One rank of tests/test_slabs.py::test_rendezvous_failure_reaches_every_rank: the RCCL-id rendezvous and the ranks' joint decision, over gloo on CPU."""
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
if mode == "rank0_fails":
    def boom():
        raise OSError("librccl.so.1: cannot open shared object file")
    parallel.comm_unique_id = boom                       # (only rank 0 calls it)
else:
    parallel.comm_unique_id = lambda: b"\x07" * 128
err, uid = None, None
try:
    uid = parallel.rendezvous_unique_id(rank)
except Exception as e:                                   # noqa: BLE001
    err = f"{type(e).__name__}: {e}"
if mode == "rank1_fails_later" and rank == 1:
    err = "SmacError: smac_comm_init failed"
failed = parallel.all_ranks_ok(err)
json.dump({"uid_ok": uid == b"\x07" * 128, "err": err, "failed": failed}, open(os.path.join(out, f"rank{rank}.json"), "w"))
dist.barrier()
dist.destroy_process_group()
