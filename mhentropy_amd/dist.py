"""One process per GPU; the forward+loss path shards on images and needs no data-path
collective.  These helpers hold the rank bookkeeping that bench.py and the
harness use: the rank's image shard and barrier-bracketed, max-over-ranks timing.  Backend "nccl"
is RCCL on ROCm (xGMI); the CPU tests drive the same code over "gloo"."""
import os
import time

import torch


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise torch.distributed from the launcher's environment; returns (rank, local_rank, world, dist|None)."""
    rank, local_rank, world = env_rank()
    if world == 1:
        return rank, local_rank, world, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return rank, local_rank, world, dist


def shard_range(global_batch, rank, world):
    """Images [lo, hi) of a global batch owned by `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def timed_region(fn, steps, dist=None, device=None, sync=None):
    """barrier + sync, `steps` calls of fn, sync + barrier; returns the MAX over ranks of the elapsed seconds."""
    sync = sync or (torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None))
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def reduce_mean_scalars(values, dist=None, device=None):
    """Average a dict of python floats across ranks (logging only; never on the data path)."""
    if dist is None:
        return dict(values)
    keys = sorted(values)
    t = torch.tensor([values[k] for k in keys], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t /= dist.get_world_size()
    return {k: float(v) for k, v in zip(keys, t.tolist())}


def allreduce_gradients(flat, dist=None, bucket_elems=16 << 20):
    """The train step's ONE exchange: sum the flat gradient buffer over ranks, in place,
    as a few large buckets issued back to back (64 MB of f32 each: large enough to run every xGMI link at its
    ring bandwidth, small enough that a later bucket's reduce-scatter overlaps an earlier one's all-gather).
    The mean (DDP semantics) is taken by the optimizer kernel's grad_scale = 1/world."""
    if dist is None or dist.get_world_size() == 1:
        return flat
    works = []
    for lo in range(0, flat.numel(), bucket_elems):
        works.append(dist.all_reduce(flat[lo:lo + bucket_elems], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
    return flat
