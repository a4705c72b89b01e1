"""One process per GPU; the forward+loss path shards on images and needs no data-path
collective.  These helpers hold the rank bookkeeping that bench.py and the
harness use: the rank's image shard and barrier-bracketed, max-over-ranks timing.  Backend "nccl"
is RCCL on ROCm (xGMI); the CPU tests drive the same code over "gloo"."""
import os
import time

import torch


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def forced():
    """MHE_DIST_FORCE=1: run every collective even in a group of ONE rank (the one-GPU rehearsal of the RCCL code path:
    communicator set-up, stream ordering against the compute / capture streams, reduce_scatter_tensor; tests/test_gpu_rccl.py)"""
    return os.environ.get("MHE_DIST_FORCE", "0") == "1"


def init(backend=None):
    """Initialise torch.distributed from the launcher's environment; returns (rank, local_rank, world, dist|None)."""
    rank, local_rank, world = env_rank()
    if world == 1 and not forced():
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
    if dist is None or (dist.get_world_size() == 1 and not forced()):
        return flat
    works = []
    for lo in range(0, flat.numel(), bucket_elems):
        works.append(dist.all_reduce(flat[lo:lo + bucket_elems], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
    return flat


class HypothesisShards:
    """Hypothesis-sharded exchange around the image-sharded encoder ("optional hypothesis sharding";
    config C4).  Given the conditioning feature, an image's K hypotheses are independent, so once the encoder has run on each
    rank's OWN images:
      * gather_rows():    all-gather of the per-image rows (conditioning feature (B,512), targets) -> every rank holds all
                          world*B images - at most a few MB;
      * hypotheses():     the rank's contiguous slice [lo, hi) of the K hypotheses, which it evaluates for EVERY image
                          (flow + decode + likelihood on (hi-lo) * world * B rows);
      * reduce_images():  per-image partial sums over the local hypotheses -> all-reduce (sum): the means over all K;
      * scatter_grad():   dL/dfeat for all images from the local hypotheses -> reduce-scatter (sum): each rank receives the
                          gradient rows of its own images and continues its encoder's reverse pass.
    Parameter gradients of everything after the encoder are partial sums over the local hypotheses; the train step's ordinary
    flat-gradient all-reduce (sum over ranks, / world) completes them exactly as it completes the image-sharded ones.
    Worth it when flow + decode cost dominates the encoder (SMPL-size LBS, K = 128), not for the hand model at K = 64.
    Absent in the reference (no torch.distributed call site)."""

    def __init__(self, dist, K):
        self.dist, self.K = dist, K
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.active = dist is not None and (self.world > 1 or forced())       # collectives are issued (a forced group of one included)

    def hypotheses(self):
        return shard_range(self.K, self.rank, self.world)

    def gather_rows(self, t):
        """(B, ...) per rank -> (world*B, ...) on every rank, rank-major"""
        if not self.active:
            return t
        t = t.contiguous()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        self.dist.all_gather_into_tensor(out, t)
        return out

    def gather_hypothesis_rows(self, t, B):
        """sample-major rows (K*B, ...) of the rank's own images -> this rank's hypotheses of ALL images, sample-major over the
        gathered batch: ((hi-lo) * world*B, ...).  (Only parity runs ship host noise this way; a production run draws the
        slice's base noise on the device.)"""
        if not self.active:
            return t
        K = self.K
        full = self.gather_rows(t.reshape(K, B, *t.shape[1:]).transpose(0, 1).contiguous())        # (world*B, K, ...)
        lo, hi = self.hypotheses()
        return full[:, lo:hi].transpose(0, 1).reshape((hi - lo) * full.shape[0], *t.shape[1:]).contiguous()

    def reduce_images(self, t):
        """sum over ranks of a per-image tensor (in place)"""
        if self.active:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def scatter_grad(self, g_all):
        """(world*B, F) partial gradients -> (B, F): the sum over ranks of the rows of this rank's own images"""
        if not self.active:
            return g_all
        B = g_all.shape[0] // self.world
        g_all = g_all.contiguous()
        if self.dist.get_backend() == "nccl":               # RCCL: one reduce-scatter, every xGMI link carries 1/world of it
            out = torch.empty((B,) + tuple(g_all.shape[1:]), device=g_all.device, dtype=g_all.dtype)
            self.dist.reduce_scatter_tensor(out, g_all, op=self.dist.ReduceOp.SUM)
            return out
        self.dist.all_reduce(g_all, op=self.dist.ReduceOp.SUM)       # gloo (CPU tests, one-GPU rehearsal) has no reduce-scatter
        return g_all[self.rank * B:(self.rank + 1) * B].contiguous()
