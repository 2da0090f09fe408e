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


def _pack_buckets(params, bucket_elems):
    """``.grad`` of ``params`` gathered into flat f32 buckets: [(flat, views of flat shaped like the grads, the parameters)].  One
    concatenation kernel per 128 tensors on the caller's stream (capturable).  (As `torch._foreach_copy_` the same 130 MB took 14
    multi-tensor launches and 0.46 ms per iteration, in and out: 0.3 TB/s on lists that mix 4 M-element weights with 64-element biases.)"""
    owners = [p for p in params if p.grad is not None]
    out, bucket, n = [], [], 0

    def flush():
        nonlocal bucket, n
        if not bucket:
            return
        g0 = bucket[0].grad
        flat = torch.empty(n, dtype=g0.dtype, device=g0.device)
        torch.cat([p.grad.reshape(-1) for p in bucket], out=flat)
        parts = [v.view_as(p.grad) for v, p in zip(flat.split([p.grad.numel() for p in bucket]), bucket)]
        out.append((flat, parts, bucket))
        bucket, n = [], 0

    for p in owners:
        if n + p.grad.numel() > bucket_elems:
            flush()
        bucket.append(p)
        n += p.grad.numel()
    flush()
    return out


def _mean_op():
    """(reduce op, whether the sum still has to be divided): RCCL averages inside the collective, gloo has no such op"""
    return (dist.ReduceOp.AVG, False) if dist.get_backend() == "nccl" else (dist.ReduceOp.SUM, True)


def _adopt_buckets(packed, W, divide):
    """The reduced buckets BECOME the gradients: every ``p.grad`` is re-pointed at its slice of the flat buffer (nothing is copied back;
    the optimizer reads the slices)."""
    for flat, parts, bucket in packed:
        if divide:
            flat.mul_(1.0 / W)
        for p, v in zip(bucket, parts):
            p.grad = v


@torch.no_grad()
def allreduce_mean_grads(params, bucket_elems=64 * 1024 * 1024):
    """Average ``.grad`` over ranks in flat f32 buckets (<= 256 MB each: D's 81 MB and G's 49 MB of gradients at 256 px are ONE
    collective each -- the 8 GPUs of a node are fully connected over xGMI, and a ring all-reduce of B bytes moves 2*(N-1)/N*B
    per link pair whatever the bucket count, so fewer, larger collectives only save launch latency).  Parameters whose grad is
    None are skipped; every rank runs the same graph so the skip pattern is identical.
    The gather into the bucket is a kernel on the caller's stream (capturable) and the reduced bucket becomes the gradients (each ``.grad``
    is re-pointed at its slice); only the collective itself is an eager seam of a captured iteration (graph.seam)."""
    if not active():
        return
    from . import graph
    packed = _pack_buckets(params, bucket_elems)
    op, divide = _mean_op()
    for flat, _, _ in packed:
        graph.seam(lambda flat=flat: dist.all_reduce(flat, op=op))
    _adopt_buckets(packed, world(), divide)


@torch.no_grad()
def allreduce_mean_grads_begin(params, bucket_elems=64 * 1024 * 1024):
    """First half of an all-reduce that runs BESIDE the rest of a backward pass: the gradients of ``params`` (those the backward has
    finished with: the discriminator's last blocks and head, 93 % of its gradient bytes, done after about a quarter of its backward time)
    are packed and their collective is started without waiting for it -- on RCCL it runs on the process group's own stream while the
    caller's stream goes on with the remaining backward.  Returns the handle for ``allreduce_mean_grads_end``; None when the step issues
    no collectives.  Under graph capture the start is one eager seam (graph.seam), like the collective of ``allreduce_mean_grads``."""
    if not active():
        return None
    from . import graph
    packed = _pack_buckets(params, bucket_elems)
    works = [None] * len(packed)
    op, _ = _mean_op()

    def start():
        for k, (flat, _, _) in enumerate(packed):
            works[k] = dist.all_reduce(flat, op=op, async_op=True)
    graph.seam(start)
    return packed, works


@torch.no_grad()
def allreduce_mean_grads_end(handle, params=(), bucket_elems=64 * 1024 * 1024):
    """Second half: the gradients of ``params`` (what the rest of the backward produced) are all-reduced, the collectives started by
    ``allreduce_mean_grads_begin`` are waited for (the caller's stream waits, not the host), and every gradient is scaled to the mean
    (inside the collective on RCCL) and ``.grad`` of every parameter is re-pointed at its slice of the reduced buckets.  One eager seam under capture."""
    if handle is None:
        return allreduce_mean_grads(params, bucket_elems)
    from . import graph
    packed, works = handle
    rest = _pack_buckets(params, bucket_elems)
    op, divide = _mean_op()

    def finish():
        for flat, _, _ in rest:
            dist.all_reduce(flat, op=op)
        for w in works:
            w.wait()
    graph.seam(finish)
    _adopt_buckets(packed + rest, world(), divide)


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


def gather_rows_multi(tensors):
    """``gather_rows`` of several [n, ...] tensors in ONE collective (they are concatenated along their flattened feature axis, gathered,
    and split again): every collective is a seam of the captured iteration, and the contrastive terms of a step all-gather their image
    embeddings, text embeddings and pooled features at the same point.  Differentiable; the tensors themselves when not distributed."""
    tensors = list(tensors)
    if not active() or len(tensors) < 2:
        return [gather_rows(t) for t in tensors]
    n = tensors[0].shape[0]
    flat = [t.reshape(n, -1).float() for t in tensors]
    g = _GatherRows.apply(torch.cat(flat, 1))
    outs = g.split([f.shape[1] for f in flat], 1)
    return [o.contiguous().to(t.dtype).reshape((g.shape[0],) + tuple(t.shape[1:])) for o, t in zip(outs, tensors)]

