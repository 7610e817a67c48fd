"""Page-level sharding across the GPUs of one node.

Pages are independent calls in the reference (src/binarizations/binarizeSauvola.cpp:32-134 touches only its
two Mats), so the multi-GPU path is a contiguous block split of the page list: one process per GPU, no
data-path collective.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for the
barrier around the timed region and for the max-over-ranks of the elapsed time.
"""
from __future__ import annotations

import os


def page_range(n_pages: int, world: int, rank: int) -> range:
    """Contiguous block of the page list owned by `rank` (sizes differ by at most one page)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(n_pages, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str | None = None):
    """Join the process group described by torchrun's environment (no-op for a single process)."""
    import torch
    import torch.distributed as dist

    world, rank, local_rank = env_world()
    on_gpu = torch.cuda.is_available()
    if on_gpu:
        # one process per GPU: bind this process to its device BEFORE the first collective (RCCL communicators and
        # barrier() are created on the current device; every rank left on device 0 is a "duplicate GPU" error)
        torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    # (PRL_FORCE_DIST=1: join a process group even for a single rank - exercises the RCCL plumbing on one GPU)
    if (world > 1 or os.environ.get("PRL_FORCE_DIST") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PRL_DIST_BACKEND=gloo: several ranks on ONE GPU (testing the N>1 path on a single-GPU box; RCCL refuses that)
        dist.init_process_group(backend or os.environ.get("PRL_DIST_BACKEND") or ("nccl" if on_gpu else "gloo"))
    return world, rank, local_rank


def barrier():
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if dist.get_backend() != "nccl":
        device = None   # (gloo: host tensors, also when the ranks hold GPUs - PRL_DIST_BACKEND=gloo)
    elif device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if dist.get_backend() != "nccl":
        device = None   # (gloo: host tensors, also when the ranks hold GPUs - PRL_DIST_BACKEND=gloo)
    elif device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def backend_name():
    import torch.distributed as dist

    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None


def gather_lists(items: list) -> list:
    """Concatenation over the ranks (in rank order) of each rank's list of small picklable items."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return list(items)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, list(items))
    return [x for part in parts for x in part]


def finish():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        barrier()
        dist.destroy_process_group()
