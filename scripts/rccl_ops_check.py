"""The collectives bench.py issues, on RCCL with the ranks the launcher gives (one rank on a one-GPU box):
uint8 all_gather_into_tensor of tile buffers, float64 all_reduce SUM / MAX, barrier.
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 scripts/rccl_ops_check.py
"""
import os

import torch
import torch.distributed as dist

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
device = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
part = torch.full((4096 * 768,), rank + 1, dtype=torch.uint8, device=device)
whole = torch.zeros(world * part.numel(), dtype=torch.uint8, device=device)
dist.all_gather_into_tensor(whole, part)
done = torch.cuda.Event()
done.record()
done.synchronize()
assert all(int(whole[r * part.numel()]) == r + 1 and int(whole[(r + 1) * part.numel() - 1]) == r + 1 for r in range(world))
t = torch.tensor([1.5 + rank, 2.0], dtype=torch.float64, device=device)
m = t.clone()
dist.all_reduce(t, op=dist.ReduceOp.SUM)
dist.all_reduce(m, op=dist.ReduceOp.MAX)
assert t.tolist() == [sum(1.5 + r for r in range(world)), 2.0 * world] and m.tolist() == [1.5 + world - 1, 2.0]
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
if rank == 0:
    print("rccl ops ok, world", world)
