"""Data-parallel host logic on CPU with world_size 2 (gloo): the N-rank step on shards must equal the 1-rank step on
the concatenated batch.

The collective plumbing under test is the product's (`xmc_gan_amd.parallel`: mean gradient all-reduce in flat buckets and
the differentiable row all-gather whose backward is scaled by world_size); the per-rank arithmetic is supplied by the
CPU oracle, because the product's kernels need a GPU.
"""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _loss_terms(X, h, PD, PG, batch, gather):
    """D-step style loss on one shard: batch-mean hinge terms + contrastive term over (gathered) embeddings."""
    import torch.nn.functional as F
    psent = X.proj_sent(PG, batch["sent_embs"])
    feat = X.netd_forward(PD, h, batch["imgs"])
    logit, img_emb, txt_emb = X.cond_dnet(PD, h, feat, psent)
    hinge = F.relu(1.0 - logit).mean()
    ie, te = gather(img_emb), gather(txt_emb)
    labels = X.make_labels(ie.size(0), None, False)
    return hinge + X.contrastive_loss(ie, te, labels, False)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import xmc_ref as X
    from xmc_gan_amd import parallel
    h = X.Hyper(img_size=64, nch=8)
    PG = X.synth_params(X.netg_shapes(h), 1)
    PD = {k: v.clone().requires_grad_() for k, v in X.synth_params(X.netd_shapes(h), 2).items()}
    full = X.synth_batch(h, 6, seed=9)
    n = 6 // world
    shard = {k: v[rank * n:(rank + 1) * n] for k, v in full.items()}
    loss = _loss_terms(X, h, PD, PG, shard, parallel.gather_rows)
    loss.backward()
    params = [p for p in PD.values()]
    # one parameter without a gradient on every rank (like the unused conv_s at 256 px) must be skipped consistently
    assert PD["downblocks.0.conv_s.weight"].grad is not None
    extra = torch.nn.Parameter(torch.zeros(3))
    parallel.allreduce_mean_grads(params + [extra], bucket_elems=5000)      # small buckets -> several collectives
    assert extra.grad is None
    if rank == 0:
        torch.save({k: v.grad.clone() for k, v in PD.items()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_on_concatenated_batch():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import xmc_ref as X
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out_path = os.path.join(tempfile.mkdtemp(), "grads.pt")
    procs = [ctx.Process(target=_worker, args=(r, world, port, out_path)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    got = torch.load(out_path)
    h = X.Hyper(img_size=64, nch=8)
    PG = X.synth_params(X.netg_shapes(h), 1)
    PD = {k: v.clone().requires_grad_() for k, v in X.synth_params(X.netd_shapes(h), 2).items()}
    full = X.synth_batch(h, 6, seed=9)
    loss = _loss_terms(X, h, PD, PG, full, lambda t: t)
    loss.backward()
    for k, v in PD.items():
        if v.grad is None:
            assert k not in got or got[k] is None
            continue
        torch.testing.assert_close(got[k], v.grad, rtol=2e-4, atol=1e-6, msg=lambda m: f"{k}: {m}")


def test_gather_rows_identity_without_process_group():
    sys.path.insert(0, ROOT)
    from xmc_gan_amd import parallel
    x = torch.randn(4, 3, requires_grad=True)
    assert parallel.gather_rows(x) is x and parallel.world() == 1 and parallel.rank() == 0
    parallel.allreduce_mean_grads([x])      # no-op


def _toy(seed=3):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g) * 0.3) for s in ((16, 8), (16,), (12, 16), (12,), (1, 12), (5,))]     # last: never used


def _toy_loss(P, x, cut_out=None):
    h1 = torch.relu(x @ P[0].t() + P[1])
    if cut_out is not None:
        cut_out.append(h1)
    h2 = torch.relu(h1 @ P[2].t() + P[3])
    return (h2 @ P[4].t()).pow(2).mean()


def _worker_two_part(rank, world, port, out_path):
    """the discriminator step's two-part backward (xmc_gan/train_gan.py): gradients of the layers after a cut first, their all-reduce
    started without waiting, the rest of the backward, then the second half -- on a three-layer stand-in"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from xmc_gan_amd import parallel
    P = _toy()
    x = torch.randn(8, 8, generator=torch.Generator().manual_seed(5))
    n = 8 // world
    cuts = []
    loss = _toy_loss(P, x[rank * n:(rank + 1) * n], cuts)
    late, early = P[2:], P[:2]
    gs = torch.autograd.grad(loss, cuts + late, allow_unused=True)
    for p_, g_ in zip(late, gs[len(cuts):]):
        p_.grad = g_
    assert P[5].grad is None and P[0].grad is None
    pending = parallel.allreduce_mean_grads_begin(late, bucket_elems=100)          # several buckets in flight
    assert pending is not None and len(pending[0]) >= 2
    torch.autograd.backward(cuts, list(gs[:len(cuts)]))
    parallel.allreduce_mean_grads_end(pending, early, bucket_elems=100)
    if rank == 0:
        torch.save([None if p_.grad is None else p_.grad.clone() for p_ in P], out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_part_backward_with_the_first_all_reduce_in_flight_equals_one_backward_on_the_whole_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out_path = os.path.join(tempfile.mkdtemp(), "grads2.pt")
    procs = [ctx.Process(target=_worker_two_part, args=(r, world, port, out_path)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    got = torch.load(out_path)
    P = _toy()
    x = torch.randn(8, 8, generator=torch.Generator().manual_seed(5))
    # mean over ranks of the shard means == the mean over the whole batch (equal shards)
    _toy_loss(P, x).backward()
    assert got[5] is None and P[5].grad is None
    for k in range(5):
        torch.testing.assert_close(got[k], P[k].grad, rtol=1e-5, atol=1e-7, msg=lambda m: f"parameter {k}: {m}")


def test_two_part_all_reduce_is_a_no_op_without_a_process_group():
    sys.path.insert(0, ROOT)
    from xmc_gan_amd import parallel
    x = torch.nn.Parameter(torch.randn(4, 3))
    x.grad = torch.ones_like(x)
    assert parallel.allreduce_mean_grads_begin([x]) is None
    parallel.allreduce_mean_grads_end(None, [x])
    assert torch.equal(x.grad, torch.ones_like(x))
