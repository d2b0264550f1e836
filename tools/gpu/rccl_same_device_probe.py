#!/usr/bin/env python3
"""GPU box helper: can two ranks on ONE device form an RCCL communicator?  (torch.distributed.run --nproc-per-node 2)"""
import os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda") * (rank + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce over RCCL on one shared device -> {t.tolist()}", flush=True)
    gl = [torch.empty(4, device="cuda") for _ in range(world)] if rank == 0 else None
    dist.gather(t, gl, dst=0)
    torch.cuda.synchronize()
    print(f"rank {rank}: gather ok", flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}: RCCL refused: {type(e).__name__}: {str(e)[:300]}", flush=True)
    sys.exit(3)
