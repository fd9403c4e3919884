"""RCCL on the one GPU of the box: a `nccl` process group of ONE rank bound to cuda:0 (the form bench.py --gpus N uses per rank),
sharding.RowGather's two collectives (all_gather_into_tensor for equal shards; the padded all_gather form is forced too) and
an all_reduce(MAX) like bench.py's timing reduction.  Degenerate (nothing crosses a link) but it is RCCL's own init, its
communicator on the device and its kernels that run."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from nav_gym_amd import sharding
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
t0 = time.perf_counter()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
obs = torch.arange(4096 * 1088, dtype=torch.float32, device=dev).reshape(4096, 1088)
g = sharding.RowGather(4096, (1088,), torch.float32, dev, 0, 1)
out = g.run(obs)
torch.cuda.synchronize()
assert torch.equal(out, obs)
t1 = time.perf_counter()
for _ in range(20):
    g.run(obs)
torch.cuda.synchronize()
per = (time.perf_counter() - t1) / 20
g2 = sharding.RowGather(4096, (1088,), torch.float32, dev, 0, 1)
g2.equal = False; g2.pad = torch.zeros_like(obs); g2.parts = [torch.empty_like(obs)]
assert torch.equal(g2.run(obs), obs)
x = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(x, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("nccl (RCCL) process group of one rank on %s: init + first gather %.2f s; all_gather_into_tensor of 4096 x 1088 f32 "
      "(17.8 MB) %.1f us per call; padded all_gather and all_reduce(MAX) ok; backend %s, version %s"
      % (torch.cuda.get_device_name(0), t1 - t0, per * 1e6, dist.get_backend(), ".".join(map(str, torch.cuda.nccl.version()))))
dist.destroy_process_group()
