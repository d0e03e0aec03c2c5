#!/usr/bin/env python3
"""The collectives bench.py makes between its ranks when N > 1 (an RCCL process group bound to the rank's device, a barrier, a MAX all-reduce of one double), in a world of one: checks that they work in this image on a box with one GPU."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
torch.cuda.synchronize(dev); dist.barrier(); torch.cuda.synchronize(dev)
t = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("nccl world of one: barrier and MAX all-reduce ok", float(t.item()))
dist.destroy_process_group()
