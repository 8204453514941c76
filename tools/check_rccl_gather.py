#!/usr/bin/env python3
"""On a GPU box: the RCCL calls of the sharded path with the only world size one GPU allows (1): process-group init as bench.py does
it, dist.gather into VIEWS of one buffer on a side stream that waits for the producer stream (what ShardedFrame._gather does),
all_reduce of the timing tensor, barrier.  Checks that the gathered bytes land in the shared buffer."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
n = 1012 * 256
local = torch.arange(n, dtype=torch.int32, device=dev).reshape(n, 1)
all8 = torch.zeros((1, n, 1), dtype=torch.int32, device=dev)
parts = [all8[r] for r in range(1)]
producer = torch.cuda.Stream(device=dev)
comm = torch.cuda.Stream(device=dev, priority=-1)
with torch.cuda.stream(producer):
    local.mul_(3)
comm.wait_stream(producer)
with torch.cuda.stream(comm):
    dist.gather(local, gather_list=parts, dst=0)
comm.synchronize()
assert parts[0].data_ptr() == all8.data_ptr()
assert torch.equal(all8[0, :, 0].cpu(), (torch.arange(n, dtype=torch.int32) * 3)), "gather did not write into the shared buffer"
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("rccl gather into views on a side stream: ok; all_reduce:", float(t.item()))
dist.destroy_process_group()
