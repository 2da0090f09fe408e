"""Data-parallel plumbing for the G+D step: one process per GPU, RCCL (torch.distributed 'nccl') over xGMI.

The reference is single-GPU (train_gan.py:427); data parallelism is defined here by equivalence with the
single-process step on the concatenated batch:
  * batch-mean losses (hinge terms, MA-GP)  ->  gradient MEAN all-reduce over ranks;
  * batch-coupled contrastive terms          ->  every rank evaluates the loss on the all-gathered embeddings
    (identical value on all ranks); ``gather_rows`` scales its backward by world_size so that the mean
    all-reduce reproduces the sum over ranks of the per-rank partial derivatives.
Works on CPU tensors with the gloo backend too (used by the world_size-2 tests).
"""
import torch
import torch.distributed as dist


_FORCE = [False]


def force_collectives(on=True):
    """Run the data-parallel path (flat-bucket gradient all-reduce, embedding all-gather, graph seams) even at world size 1,
    provided a process group exists: the single-card rehearsal of the RCCL calls (bench.py --force_dp)."""
    _FORCE[0] = bool(on)


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active():
    """does the iteration issue collectives?  (through world(), so that a caller which narrows the step to one rank by
    replacing world / rank -- tests/dp_rehearsal.py's single-process references inside a 2-rank job -- also switches them off)"""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return world() > 1 or _FORCE[0]


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


@torch.no_grad()
def allreduce_mean_grads(params, bucket_elems=64 * 1024 * 1024):
    """Average ``.grad`` over ranks in flat f32 buckets (<= 256 MB each: D's 81 MB and G's 49 MB of gradients at 256 px are ONE
    collective each -- the 8 GPUs of a node are fully connected over xGMI, and a ring all-reduce of B bytes moves 2*(N-1)/N*B
    per link pair whatever the bucket count, so fewer, larger collectives only save launch latency).  Parameters whose grad is
    None are skipped; every rank runs the same graph so the skip pattern is identical.
    The pack / unpack copies are multi-tensor kernels on the caller's stream (capturable); only the collective itself is an
    eager seam of a captured iteration (graph.seam)."""
    W = world()
    if not active():
        return
    from . import graph
    grads = [p.grad for p in params if p.grad is not None]
    bucket, n = [], 0

    def flush():
        nonlocal bucket, n
        if not bucket:
            return
        flat = torch.empty(n, dtype=bucket[0].dtype, device=bucket[0].device)
        parts = [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in bucket]), bucket)]
        torch._foreach_copy_(parts, bucket)                     # one multi-tensor copy in ...
        graph.seam(lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM))
        flat.mul_(1.0 / W)
        torch._foreach_copy_(bucket, parts)                     # ... and one back
        bucket, n = [], 0

    for g in grads:
        if n + g.numel() > bucket_elems:
            flush()
        bucket.append(g)
        n += g.numel()
    flush()


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        W = world()
        x = x.contiguous()
        out = torch.empty((W * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        from . import graph
        graph.seam(lambda: dist.all_gather_into_tensor(out, x))
        ctx.n = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        r, W = rank(), world()
        return g[r * ctx.n:(r + 1) * ctx.n] * float(W)


def gather_rows(x):
    """[n, ...] -> [world*n, ...] (rank-major), differentiable; identity when not distributed."""
    return _GatherRows.apply(x) if active() else x
