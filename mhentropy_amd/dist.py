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


class GradExchange:
    """The train step's ONE exchange, bucket by bucket: `start(flat_slice)` hands a completed range of the flat f32
    gradient buffer to the communicator while the reverse pass goes on, `finish()` makes the compute stream wait for all of them.

    * EVENT-SCOPED: each bucket's collective depends on ONE event - recorded on the compute stream right after the bucket's gather - and runs
      on a communication stream of its own (backend "nccl" = RCCL).  Whatever the reverse pass queues afterwards on the compute stream is
      not in the collective's way, and the collective is not in the way of the compute stream until finish().
    * mode "f32" (default): all_reduce(SUM) of the f32 range, in place.
    * mode "bf16" (MHE_GRAD_EXCHANGE=bf16; absent in the reference, which has no distributed path at all): half the bytes on the wire WITH f32
      accumulation - an all-to-all of bf16 chunks (rank r receives chunk r of every rank: a direct reduce-scatter in which all seven xGMI
      links of a GPU carry 1/8 of the message at once), the chunks summed locally in f32 in rank order (a fixed order:
      every rank computes the same sums), the reduced chunk rounded to bf16 and all-gathered.  Every rank ends with the SAME bits, so
      replicas stay identical; the gradient carries one bf16 rounding per element and rank before the sum and one after it.
    The mean over ranks (DDP semantics) is taken by the optimizer kernel's grad_scale = 1 / world in both modes."""

    def __init__(self, dist, mode=None, device=None):
        self.dist, self.mode = dist, (mode or os.environ.get("MHE_GRAD_EXCHANGE", "f32"))
        if self.mode not in ("f32", "bf16"):
            raise ValueError(f"gradient exchange mode {self.mode!r}: 'f32' or 'bf16'")
        self.world = dist.get_world_size()
        self.side = torch.cuda.Stream(device=device) if (device is not None and torch.device(device).type == "cuda" and dist.get_backend() == "nccl") else None
        self._pending, self._bufs = [], {}

    def _buf(self, key, n, dtype, device):
        b = self._bufs.get(key)
        if b is None or b.numel() < n or b.dtype != dtype:
            b = self._bufs[key] = torch.empty(n, dtype=dtype, device=device)
        return b[:n]

    def _exchange(self, g, key):
        """the collective(s) of one bucket, issued on the current stream context; returns the work handle(s) to wait on"""
        d = self.dist
        if self.mode == "f32":
            return [d.all_reduce(g, op=d.ReduceOp.SUM, async_op=True)]
        W, n = self.world, g.numel()
        c = -(-n // (8 * W)) * 8                                    # chunk length: a multiple of 8 elements (16-byte bf16 pieces)
        send = self._buf((key, "send"), W * c, torch.bfloat16, g.device)
        send[:n].copy_(g)
        if W * c > n:
            send[n:].zero_()
        recv = self._buf((key, "recv"), W * c, torch.bfloat16, g.device)
        w1 = d.all_to_all_single(recv, send, async_op=True)
        w1.wait()                                                   # (stream-ordered on the device: the host does not block on nccl; gloo: it does)
        red = recv.view(W, c).float().sum(0).to(torch.bfloat16)     # f32 accumulation, rank order
        out = self._buf((key, "out"), W * c, torch.bfloat16, g.device)
        w2 = d.all_gather_into_tensor(out, red, async_op=True)
        w2.wait()
        g.copy_(out[:n])
        return []

    def start(self, g, key=0):
        if self.side is None:                                       # gloo (CPU tests, one-GPU rehearsals): the backend's own ordering
            self._pending += self._exchange(g, key)
            return
        ev = torch.cuda.Event()
        ev.record()                                                 # the bucket's gather has been queued on the compute stream: depend on exactly that
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            works = self._exchange(g, key)
            for w in works:
                w.wait()                                            # the side stream waits for the communicator's stream (no host block)
            done = torch.cuda.Event()
            done.record()
        g.record_stream(self.side)
        self._pending.append(done)

    def in_flight(self):
        return len(self._pending)

    def finish(self):
        for p in self._pending:
            if isinstance(p, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(p)
            else:
                p.wait()
        self._pending = []


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
