"""Batch-sharded data parallelism for the train step: one process per GPU, RCCL over xGMI
(torch.distributed backend "nccl" is RCCL on ROCm; "gloo" on CPU for the tests).

The reference is single-process (no collective anywhere); what its semantics imply once the batch is
sharded is implemented here:
  1. class weights / CE normaliser are functions of the GLOBAL batch -> all-reduce of (positives, rows)
     before the loss; every rank then scales its local numerator by the global denominator and the
     gradients are SUMMED, not averaged;
  2. one bucketed all-reduce of all gradients per step (a few MB: latency-, not bandwidth-bound on the
     fully connected 8-GPU xGMI mesh, so ONE flat bucket);
  3. the modality-dropout gate and every selection uniform are keyed on (seed, step, GLOBAL row), so
     the sampled index tensor does not depend on the number of ranks;
  4. BatchNorm: local statistics by default (as torch DDP); `set_sync_batchnorm(model)` switches the sequence
     pre-network to statistics of the GLOBAL batch (the single-process result) -- SURVEY 8e(2).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for world size 1)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard_rows(global_batch, rank_, world):
    """Contiguous row block of rank_: (row0, rows).  Remainder rows go to the lowest ranks."""
    base, rem = divmod(int(global_batch), int(world))
    rows = base + (1 if rank_ < rem else 0)
    row0 = rank_ * base + min(rank_, rem)
    return row0, rows


def allreduce_counts(class_counts):
    """(positives, rows) of the local shard -> of the global batch, in place (SUM)."""
    if world_size() > 1:
        dist.all_reduce(class_counts, op=dist.ReduceOp.SUM)
    return class_counts


FORCE_COLLECTIVES = False    # rehearsal: issue the collectives with a single rank too (bench.py --force-collectives)


def collectives_on():
    return world_size() > 1 or (FORCE_COLLECTIVES and dist.is_initialized())


def allreduce_sum_(t):
    """SUM a device vector over the ranks in place (BatchNorm sums of the global batch, functional._ConvStackFn)."""
    if collectives_on():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def set_sync_batchnorm(model, enable=True):
    """Parity switch for data-parallel runs: every CNN_pre inside `model` takes its BatchNorm statistics (and the two
    means of the BatchNorm backward) over the global batch instead of the rank's shard."""
    n = 0
    for m in model.modules():
        if hasattr(m, "sync_batchnorm"):
            m.sync_batchnorm = bool(enable)
            n += 1
    return n


class GradBucket:
    """One flat all-reduce for all gradients of a parameter list."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.flat = None

    def allreduce(self):
        """SUM the gradients over ranks (the loss already carries the global normaliser)."""
        if world_size() == 1:
            return
        grads = [p.grad for p in self.params if p.grad is not None]
        if not grads:
            return
        n = sum(g.numel() for g in grads)
        if self.flat is None or self.flat.numel() != n or self.flat.dtype != grads[0].dtype \
                or self.flat.device != grads[0].device:
            self.flat = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
        off = 0
        for g in grads:
            self.flat[off:off + g.numel()].copy_(g.reshape(-1))
            off += g.numel()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        off = 0
        for g in grads:
            g.copy_(self.flat[off:off + g.numel()].view_as(g))
            off += g.numel()


class FlatGrads:
    """All gradients of a parameter list live in ONE flat buffer: `p.grad` is a view into it and the backward
    kernels write there directly (functional.register_grad_sink), so the data-parallel reduction is a single
    in-place all-reduce with no flatten / unflatten copies.  `extra` trailing slots ride along in the same
    collective (bench.py uses them to pre-reduce the next step's class counts).  Do not call
    zero_grad(set_to_none=True) on these parameters."""

    def __init__(self, params, extra=0):
        from . import functional as F_
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dt, dev = self.params[0].dtype, self.params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]        # keep every view 16-byte aligned
        self.flat = torch.zeros(sum(sizes) + extra, dtype=dt, device=dev)
        off = 0
        for p, n in zip(self.params, sizes):
            if p.dtype != dt or p.device != dev:
                raise TypeError("FlatGrads needs parameters of one dtype on one device")
            view = self.flat[off:off + p.numel()].view_as(p)
            p.grad = view
            F_.register_grad_sink(p, view)
            off += n
        self.extra = self.flat[off:off + extra] if extra else None

    def allreduce(self):
        if dist.is_initialized():          # also with a single rank (bench.py --force-collectives rehearses the capture)
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)


class BucketedFlatGrads:
    """FlatGrads in TWO buckets so that the reduction of the first overlaps the rest of the backward pass:
    "early" = parameters whose gradients are complete once the fusion layer's backward has been enqueued (post stack,
    classifier head, docking layers), "late" = the pre-networks.  `allreduce_early()` issues the first collective
    asynchronously (RCCL runs it on its own stream behind the work enqueued so far), `finish()` waits for it and reduces the
    second bucket.  Both are capturable into a step graph with the RCCL backend."""

    def __init__(self, model, extra=0):
        early, late = [], []
        for name, p in model.named_parameters():
            if p.requires_grad:
                (early if name.startswith(("embracenet.", "post.")) else late).append(p)
        self.early = FlatGrads(early) if early else None
        self.late = FlatGrads(late, extra=extra) if (late or extra) else None   # `extra` slots ride in the LAST collective
        self.extra = self.late.extra if self.late is not None else None
        self._work = None

    @staticmethod
    def _active():
        return dist.is_initialized() and (world_size() > 1 or FORCE_COLLECTIVES)

    def buckets(self):
        return [b for b in (self.early, self.late) if b is not None]

    def allreduce_early(self):
        if self.early is not None and self._active():
            self._work = dist.all_reduce(self.early.flat, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        if not self._active():
            return
        if self._work is not None:
            self._work.wait()
            self._work = None
        elif self.early is not None:                 # the early hook did not fire (no fusion layer in the model)
            dist.all_reduce(self.early.flat, op=dist.ReduceOp.SUM)
        if self.late is not None:
            dist.all_reduce(self.late.flat, op=dist.ReduceOp.SUM)

    def zero_(self):
        for b in self.buckets():
            b.flat.zero_()


def broadcast_buffers(model, src=0):
    """Make every rank hold rank `src`'s module buffers (BatchNorm running statistics differ per rank under local
    statistics): called before a checkpoint is written / a fit returns, so that all replicas evaluate alike."""
    if world_size() > 1:
        for b in model.buffers():
            dist.broadcast(b, src)


def barrier():
    if world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
