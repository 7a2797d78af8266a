"""RCCL sanity checks of the two hosts, as far as a one-GPU box can take them:
  (1) the exact torch.distributed calls bench.py makes at N > 1 (init with device_id, barrier, all_reduce MAX, destroy);
  (2) the in-library RLS_COMM_RCCL transport (rls_comm_create -> ncclCommInitAll, one ncclAllReduce per exchange) with ONE rank:
      a row-sharded CGNR solve through rls_cgnr_*_rowsharded must equal the single-GPU solve bit for bit;
  (3) bench.py's one-process config-5 leg (multigpu.bench_rowsharded_one_process) at a reduced size with one rank.
    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/rccl_sanity.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, torch.distributed as dist
lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("dist ok", float(t.item()), dist.get_world_size())

import rls_amd as rls
from rls_amd import multigpu as mg
rng = np.random.default_rng(11)
M, N = 1024, 512
A = np.asfortranarray((rng.standard_normal((M, N)) + 1j * rng.standard_normal((M, N))).astype(np.complex64))
b = (A @ (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)).astype(np.complex64)
ref = rls.solve_(rls.createLinearSolver(rls.CGNR, rls.DeviceMatrix.from_host(A), iterations=12, relTol=0.0),
                 rls.DeviceVector.from_host(b)).to_host()
for name, tr in (("rccl", mg.COMM_RCCL), ("direct", mg.COMM_DIRECT)):
    s = mg.CommRowShardedCGNR(rls, [A], devices=[lr], transport=tr, iterations=12, relTol=0.0)
    x = s.solve([b])
    print(f"in-library transport {name}: code {s.transport}, one rank, |x - x_single| / |x| = {np.linalg.norm(x - ref) / np.linalg.norm(ref):.2e}")
    s.close()
r = mg.bench_rowsharded_one_process(rls, 0, 1, 32, 8, M=8192, N=2048)
print("one-process config-5 leg, one rank, 8192 x 2048:", {k: (round(v["iterations_per_s"]) if "iterations_per_s" in v else v) for k, v in r.items()})
dist.barrier(); dist.destroy_process_group()
