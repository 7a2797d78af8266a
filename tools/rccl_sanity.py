"""one-rank RCCL sanity check of the exact torch.distributed calls bench.py makes at N > 1 (init with device_id, barrier,
all_reduce MAX, destroy): python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/rccl_sanity.py"""
import os, torch, torch.distributed as dist
lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("dist ok", float(t.item()), dist.get_world_size())
dist.barrier(); dist.destroy_process_group()
