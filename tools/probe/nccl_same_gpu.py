"""Can two ranks share ONE GPU under RCCL on this pool?  (decides whether the in-library exchange can be exercised on a 1-GPU box)"""
import os, sys, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl")
    t = torch.full((4,), float(rank + 1), device="cuda:0")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    a = torch.arange(8, device="cuda:0", dtype=torch.float32) + 100 * rank
    b = torch.empty(8, device="cuda:0")
    peer = 1 - rank
    for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, peer), dist.P2POp(dist.irecv, b, peer)]):
        r.wait()
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce -> {t.tolist()}  recv -> {b[:3].tolist()}", flush=True)
except Exception as e:
    print(f"rank {rank}: FAILED {type(e).__name__}: {str(e)[:400]}", flush=True)
    sys.exit(3)
