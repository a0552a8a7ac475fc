"""Multi-GPU plumbing: one process per GPU (torch.distributed, backend "nccl" = RCCL on
ROCm; "gloo" in CPU tests). The path shards by film tile: rank r renders the 16x16 tiles
with tile_id % world == r for all samples (same Halton indices as a 1-GPU render), into
a full-size film that is zero outside its tiles. The only exchange is one sum-reduce of
the film at the end -- the multi-process form of Film::MergeFilmTile
(src/core/film.cpp:124-142), which the reference performs under a mutex per tile.

`ShardedFrame.step()` is the one step function: bench.py times it, tests/test_distributed.py
runs it on two gloo ranks (with the CPU oracle standing in for the device render) and
tests/test_distributed_gpu.py on two ranks with the HIP path.
"""
import os
import time


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment. Returns
    (rank, world_size, local_rank). No-op for a single process. MASTER_ADDR / MASTER_PORT come from the
    launcher (torchrun sets them); a hard-coded default port would make two jobs on one node collide."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            if "MASTER_PORT" not in os.environ:
                raise RuntimeError("WORLD_SIZE > 1 but MASTER_PORT is not set: launch with torch.distributed.run "
                                   "(--master-addr 127.0.0.1 --master-port P) or export MASTER_ADDR / MASTER_PORT")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend is None:
                import torch
                backend = os.environ.get("MIPT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_of(rank, world):
    """(shard_index, shard_count) handed to mi_pt_render / oracle_render."""
    return rank, world


class _DevicePointer:
    """A device buffer owned by the renderer, described through the CUDA array interface so that torch can
    view it without a copy."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def device_film_tensor(integ):
    """The renderer's resident film ([H, W, 32] float32: 31 bins + filter-weight sum per pixel) as a torch
    tensor on its device -- a view, not a copy: a collective on it reduces the film in place."""
    import torch
    ptr, n = integ.device_film()
    w, h = integ.scene.film_size
    assert n == w * h * 32
    return torch.as_tensor(_DevicePointer(ptr, (h, w, 32)), device="cuda")


def reduce_film(film, weight=None, dst=0):
    """Sum the per-rank films onto rank `dst` (torch tensors, in place). Tiles are owned
    by exactly one rank when the filter radius is 0.5, so the sum is exact (x + 0).
    Only rank `dst` holds the sum afterwards; what the other ranks' tensors hold is up to the backend."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return film, weight
    for t in (film, weight):
        if t is not None and not t.is_contiguous():
            raise ValueError("reduce_film needs contiguous tensors (the collective works on the storage)")
    if dist.get_backend() == "gloo" and film.is_cuda:
        # rehearsal mode (several ranks sharing one GPU, MIPT_DIST_BACKEND=gloo): reduce on the host
        for t in (film, weight):
            if t is not None:
                h = t.cpu()
                dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM)
                t.copy_(h)
        return film, weight
    dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    if weight is not None:
        dist.reduce(weight, dst=dst, op=dist.ReduceOp.SUM)
    if film.is_cuda:
        # RCCL runs on torch's communication stream and the renderer on its own HIP streams: wait on the
        # host until the reduce has read the film, so the next frame cannot overwrite it underneath.
        import torch
        torch.cuda.current_stream(film.device).synchronize()
    return film, weight


class ShardedFrame:
    """One frame of the hot path on `world` ranks: render this rank's tile shard into the film, then ONE sum-reduce
    of the film ([H, W, 32]: spectrum and filter weight of a pixel travel together) onto rank 0.

    render(shard_index, shard_count) renders into `film` (a torch tensor this object reduces in place): the HIP path
    renders into its resident device film and `film` is a view of it (device_film_tensor); a CPU test passes a
    function that fills a host tensor with the oracle's shard."""

    def __init__(self, render, film, rank, world):
        self.render, self.film, self.rank, self.world = render, film, rank, world
        self.render_s = 0.0
        self.reduce_s = 0.0
        self.steps = 0

    def step(self):
        si, sc = shard_of(self.rank, self.world)
        t0 = time.perf_counter()
        self.render(si, sc)          # blocks until the shard is in the film
        t1 = time.perf_counter()
        reduce_film(self.film, None, dst=0)
        t2 = time.perf_counter()
        self.render_s += t1 - t0
        self.reduce_s += t2 - t1
        self.steps += 1

    def per_rank_timings(self):
        """{"render_s": [...], "reduce_s": [...], "imbalance": max/mean of render_s}: seconds per step and rank. The reduce
        time of a rank includes waiting for the slowest renderer (the collective completes when all have joined)."""
        n = max(1, self.steps)
        rows = gather_rows([self.render_s / n, self.reduce_s / n])
        render = [r[0] for r in rows]
        return {"render_s": [round(v, 5) for v in render], "reduce_s": [round(r[1], 5) for r in rows],
                "imbalance": round(max(render) / max(1e-12, sum(render) / len(render)), 4)}


def broadcast_string(value, src=0):
    """rank `src`'s string on every rank (None elsewhere on entry). No-op for a single process."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    import torch
    dev = _collective_device()
    raw = value.encode() if dist.get_rank() == src else b""
    n = torch.tensor([len(raw)], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    buf = torch.zeros(int(n.item()), dtype=torch.uint8, device=dev)
    if dist.get_rank() == src:
        buf.copy_(torch.tensor(list(raw), dtype=torch.uint8))
    dist.broadcast(buf, src=src)
    return bytes(buf.cpu().tolist()).decode()


def barrier():
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            import torch
            dist.barrier(device_ids=[torch.cuda.current_device()])  # the rank's own GPU, not a guess from the rank number
        else:
            dist.barrier()


def _collective_device():
    import torch.distributed as dist
    return "cuda" if (dist.get_backend() == "nccl") else "cpu"


def max_over_ranks(value):
    """max of a python float over ranks (used for the timed region of bench.py)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_collective_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values):
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=_collective_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def gather_rows(values):
    """[[values of rank 0], [values of rank 1], ...] on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [list(values)]
    t = torch.tensor(list(values), dtype=torch.float64, device=_collective_device())
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.tolist()] for o in out]
