"""Multi-GPU plumbing: one process per GPU (torch.distributed, backend "nccl" = RCCL on
ROCm; "gloo" in CPU tests). The path shards by film tile: rank r renders the 16x16 tiles
with tile_id % world == r for all samples (same Halton indices as a 1-GPU render), into
a full-size film that is zero outside its tiles. The only exchange is one sum-reduce of
the film at the end -- the multi-process form of Film::MergeFilmTile
(src/core/film.cpp:124-142), which the reference performs under a mutex per tile.
"""
import os


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment. Returns
    (rank, world_size, local_rank). No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if backend is None:
                import torch
                backend = os.environ.get("MIPT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_of(rank, world):
    """(shard_index, shard_count) handed to mi_pt_render / oracle_render."""
    return rank, world


def reduce_film(film, weight=None, dst=0):
    """Sum the per-rank films onto rank `dst` (torch tensors, in place). Tiles are owned
    by exactly one rank when the filter radius is 0.5, so the sum is exact (x + 0)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return film, weight
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


def barrier():
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            import torch
            dist.barrier(device_ids=[torch.cuda.current_device()])  # the rank's own GPU, not a guess from the rank number
        else:
            dist.barrier()


def max_over_ranks(value):
    """max of a python float over ranks (used for the timed region of bench.py)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    dev = "cuda" if (dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values):
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(values)
    dev = "cuda" if (dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]
