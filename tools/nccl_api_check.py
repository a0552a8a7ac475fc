"""One-GPU check of the RCCL calls bench.py makes for N > 1 (world size 1, backend nccl):
init, barrier on the rank's own device, in-place reduce of a film-sized tensor, all_reduce."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t0 = time.perf_counter()
dist.barrier(device_ids=[torch.cuda.current_device()])
print("first barrier (communicator set-up) %.3f s" % (time.perf_counter() - t0))
film = torch.ones((700, 700, 31), device="cuda"); weight = torch.ones((700, 700), device="cuda")
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dist.reduce(film, dst=0, op=dist.ReduceOp.SUM); dist.reduce(weight, dst=0, op=dist.ReduceOp.SUM)
    torch.cuda.current_stream().synchronize()
    print("reduce film+weight %.3f ms" % ((time.perf_counter() - t0) * 1e3))
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(film.mean()) == 1.0 and float(t) == 1.5
dist.destroy_process_group()
print("nccl api ok")
