"""Two-rank rehearsal of the PRODUCT's data-parallel G+D iteration on real kernels (not a pytest file: started under
torch.distributed.run, see below).  Checks SURVEY.md 8e's contract on the GPU path itself:

  A1. bench cfg, plain data parallel (gradient mean all-reduce; local contrastive negatives, per-rank shifted-pair mismatch
      term): the D-phase gradients of the 2-rank step == the mean of the shards' single-process D-phase gradients.
  A2. a cfg without batch-coupled terms: the WHOLE 2-rank iteration (both Adam updates) == one single-process iteration on the
      concatenated batch: every gradient tensor and the weights afterwards.
  B.  --gather_negatives (all-gathered contrastive rows): the 2-rank iteration == ONE single-process iteration on the
      concatenated batch, contrastive losses included (RMIS_LOSS off: that term couples neighbours inside a local batch only
      and is the documented difference).
  C.  three iterations replayed as hipGraph segments with the collectives as eager seams == three eager iterations.

    XMC_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29511 tests/dp_rehearsal.py [--out profiles/r02_dp2_rehearsal.json]
  (gloo: both ranks share one card; with two cards and XMC_DIST_BACKEND=nccl the same script runs over RCCL.)
Test infrastructure: builds synthetic parameters with the oracle's generator (no oracle arithmetic is used as the reference
here -- the reference is the product's own single-process step)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p_)
import torch
import torch.distributed as dist


def run_step(h, PG, PD, batch, opts, tg, eps=1e-3):
    from parity_util import DEV, GradTap, build_product
    netG, netD, optG, optD = build_product(h, PG, PD, eps)
    tapG, tapD = GradTap(optG, netG.named_parameters()), GradTap(optD, netD.named_parameters())
    o = tg.gan_iteration(netG, netD, optG, optD, batch["imgs"].to(DEV), batch["sent_embs"].to(DEV), batch["words_embs"].to(DEV),
                         batch["mask"].to(DEV), batch["noise"].to(DEV), {}, opts)
    torch.cuda.synchronize()
    losses = {k: float(v) for k, v in o.items() if k != "fake"}
    weights = {"G." + k: v.detach().float().cpu() for k, v in netG.state_dict().items()}
    weights.update({"D." + k: v.detach().float().cpu() for k, v in netD.state_dict().items()})
    return losses, tapD.records[0], tapG.records[0], weights


def worst_rel(a, b):
    w = 0.0
    big = max([v.norm().item() for v in b.values() if v is not None] + [1e-30])
    for k, vb in b.items():
        va = a.get(k)
        assert (va is None) == (vb is None), k
        if vb is None:
            continue
        w = max(w, ((va - vb).norm() / max(vb.norm().item(), 1e-5 * big)).item())
    return w


def main():
    if os.environ.get("XMC_DUMP_AFTER"):        # where is a rank stuck?  (seconds; dumps every thread's stack and exits)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["XMC_DUMP_AFTER"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--nch", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4, help="per-rank batch")
    a = ap.parse_args()
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    backend = os.environ.get("XMC_DIST_BACKEND", "nccl")
    torch.cuda.set_device(int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count())
    dist.init_process_group(backend)
    import xmc_ref as X
    import xmc_gan.train_gan as tg
    from xmc_gan_amd import ops
    from parity_util import setup_cfg
    ops.set_precision("fp32")
    report = dict(world=world, backend=backend, per_rank_batch=a.batch, nch=a.nch, precision="fp32", cases={})
    CASES = (
        # name, yml, gather, RMIS, reference
        ("A1_bench_cfg_D_phase", "df_gan_damsm_nomagp.yml", False, True, "shard_mean_D"),
        ("A2_no_batch_coupled_terms", "df_gan_sbert_seperate.yml", False, False, "concat"),
        ("B_gather_negatives", "df_gan_damsm_nomagp.yml", True, False, "concat"),
    )
    for name, yml, gather, rmis, ref in CASES:
        cfg, h = setup_cfg(yml, **{"TRAIN.NCH": a.nch, "TRAIN.RMIS_LOSS": rmis})
        PG, PD = X.synth_params(X.gen_shapes(h), 5), X.synth_params(X.netd_shapes(h), 6)
        full = X.synth_batch(h, a.batch * world, seed=300, words_len=cfg.TEXT.MAX_LENGTH)
        shard = {k: v[rank * a.batch:(rank + 1) * a.batch] for k, v in full.items()}
        losses, gD, gG, W = run_step(h, PG, PD, shard, tg.StepOptions(gather_negatives=gather), tg)
        dist.barrier()
        if rank == 0:
            # single-process reference(s): a process group exists, so make the collectives no-ops by a world-1 view
            import xmc_gan_amd.parallel as par
            w_, r_ = par.world, par.rank
            par.world, par.rank = (lambda: 1), (lambda: 0)
            try:
                if ref == "concat":
                    rl, rD, rG, rW = run_step(h, PG, PD, full, tg.StepOptions(), tg)
                else:
                    # local negatives + per-rank mismatch pairs: the D-phase gradient of the 2-rank step is by definition the mean
                    # of the shards' single-process D-phase gradients (the later phases then start from the jointly updated D)
                    parts = [run_step(h, PG, PD, {k: v[r * a.batch:(r + 1) * a.batch] for k, v in full.items()}, tg.StepOptions(), tg)
                             for r in range(world)]
                    rD = {k: (None if parts[0][1][k] is None else sum(p[1][k] for p in parts) / world) for k in parts[0][1]}
            finally:
                par.world, par.rank = w_, r_
            res = dict(grads_D=worst_rel(gD, rD))
            if ref == "concat":
                res["grads_G"] = worst_rel(gG, rG)
                res["weights_after_step"] = worst_rel(W, rW)
                coupled = [k for k in ("ds_loss", "gs_loss", "disc_loss") if k in rl]
                if gather and coupled:      # identical on every rank: the loss over all world*B rows
                    res["contrastive_losses"] = max(abs(losses[k] - rl[k]) / (abs(rl[k]) + 1e-6) for k in coupled)
            report["cases"][name] = res
            assert all(v <= 2e-3 for v in res.values()), (name, res)
        dist.barrier()
    # C. the captured form (graph segments with the collectives as eager seams, xmc_gan_amd/graph.py) == the eager form:
    #    three iterations of the gathered-negatives configuration each way, same shards, weights compared afterwards
    from parity_util import DEV, build_product
    from xmc_gan_amd.graph import GraphedIteration
    cfg, h = setup_cfg("df_gan_damsm_nomagp.yml", **{"TRAIN.NCH": a.nch})
    PG, PD = X.synth_params(X.gen_shapes(h), 5), X.synth_params(X.netd_shapes(h), 6)
    batches = []
    for i in range(3):
        full = X.synth_batch(h, a.batch * world, seed=400 + i, words_len=cfg.TEXT.MAX_LENGTH)
        batches.append([full[k][rank * a.batch:(rank + 1) * a.batch].to(DEV) for k in ("imgs", "sent_embs", "words_embs", "mask", "noise")])
    finals = []
    for graphed in (False, True):
        netG, netD, optG, optD = build_product(h, PG, PD, 1e-3)
        opts = tg.StepOptions(gather_negatives=True)
        fn = lambda i_, s_, w_, m_, n_, st: tg.gan_iteration(netG, netD, optG, optD, i_, s_, w_, m_, n_, st, opts)
        runner = GraphedIteration(fn, batches[0], n_critic=cfg.TRAIN.N_CRITIC, warmup=1) if graphed else None
        st = {}
        for b_ in batches:
            o = runner(*b_) if graphed else fn(*b_, st)
        torch.cuda.synchronize()
        W = {"G." + k: v.detach().float().cpu().clone() for k, v in netG.state_dict().items()}
        W.update({"D." + k: v.detach().float().cpu().clone() for k, v in netD.state_dict().items()})
        finals.append((W, {k: float(v) for k, v in o.items() if k != "fake"}))
    if rank == 0:
        res = dict(weights_after_3_iterations=worst_rel(finals[1][0], finals[0][0]),
                   losses_of_iteration_3=max(abs(finals[1][1][k] - finals[0][1][k]) / (abs(finals[0][1][k]) + 1e-6) for k in finals[0][1]))
        report["cases"]["C_graph_segments_equal_eager"] = res
        assert all(v <= 2e-3 for v in res.values()), res
    dist.barrier()
    if rank == 0:
        print(json.dumps(report))
        if a.out:
            with open(a.out, "w") as f:
                json.dump(report, f, indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
