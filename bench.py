"""Benchmark of the XMC-GAN G+D training iteration on MI355X (metric and configs: BASELINE.json).

    python bench.py [--gpus N --steps K --warmup W] [--workload headline|magp|config2|config3]
                    [--imsize 256 --batch B --cfg df_gan_damsm_nomagp.yml]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full iteration of the reference loop body (train_gan.py:185-291: D step, optional MA-GP, G step,
both Adam updates) on one synthetic COCO-shaped batch that is already resident in HBM.  One process per GPU;
weak scaling (per-GPU batch fixed); value = images of all ranks / max-over-ranks time.  Rank 0 prints ONE JSON line
carrying, besides the throughput:
  roofline      the dominant kernel (MFMA view), timed per launch with HIP events on the launch stream
  roofline_hbm  the bandwidth-bound convolution kernels (algorithmic intensity below the chip's ridge): GB/s against 8 TB/s
  cpu_baseline  the CPU oracle timed on this host's cores
  parity        outside the timed region: one G+D iteration at the BENCHED image size (256x256, batch 8, the real widths NCH=32, the
                reference's own initialisation with block gammas 0.1 like the timed run) in the BENCHED precision mode against the
                f32 CPU oracle: worst relative loss error and relative L2 error of the logit vectors -- north_star's bar is 1e-3 --
                with the 64x64 figure of rounds 1-3 beside it (`at_64px`)
  alt_precision the same workload timed (with its own `roofline`) and parity-checked in the IEEE-half mode: the same kernels
                compiled for the 11-bit format, same MFMA rate, dynamic loss scale
  entrypoint    the same workload through the PRODUCT's entry point -- `xmc_gan/train_gan.py --cfg ... --synthetic N` (a child
                process): main() builds the models and the text encoder, train() feeds loader batches (host tensors, uploaded on a side
                stream) to the hipGraph replay it builds by default; images/s as train() itself reports it, beside `value`
  step_ms       median / min / max of the timed steps (`value` and `ms_per_step` are the contract's total over the K steps)
  dist          backend / world size / RCCL version the collectives ran on (`--force_dp`: the data-parallel path with its
                graph seams and RCCL calls exercised at world size 1)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p_ in (ROOT, os.path.join(ROOT, "oracle")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

import torch

# algorithmic FLOPs per image of one minimal G+D iteration = 9*D + 3*G forward-equivalents
STEP_GFLOP = {(64, False): 8.74, (128, False): 36.48, (256, False): 147.4,
              (64, True): 12.69, (128, True): 53.30, (256, True): 215.7}
D_GFLOP = {64: 0.659, 128: 2.803, 256: 11.377}           # forward GFLOP per image (measured on the oracle)
G_IN_GFLOP = {64: 1.632, 128: 6.536, 256: 26.150}         # CONCEPT_IN_DF_GEN
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0          # HBM3E spec (MI355X_MICROARCH.md; ~6300 measured with a float4 copy)


# named workloads (BASELINE.json `configs`; the default is the single-GPU leg of configs 4/5, the one `metric` is quoted on)
WORKLOADS = {
    "headline": dict(imsize=256, batch=256, cfg="df_gan_damsm_nomagp.yml"),
    "magp": dict(imsize=256, batch=256, cfg="df_gan_damsm.yml"),
    "config2": dict(imsize=64, batch=64, cfg="df_gan_damsm_nomagp.yml"),
    "config3": dict(imsize=128, batch=64, cfg="concept_in_df_gan_damsm_nomagp.yml"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", type=str, default="headline", choices=sorted(WORKLOADS),
                    help="headline = BASELINE configs 4/5 per GPU (default); magp = the same with the gradient penalty on; "
                         "config2 / config3 = BASELINE configs[1] / configs[2].  --imsize / --batch / --cfg override its parts")
    ap.add_argument("--imsize", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (BASELINE configs 4/5: 256 per GPU)")
    ap.add_argument("--cfg", type=str, default=None)
    ap.add_argument("--precision", type=str, default="bf16", choices=["bf16", "f16", "fp32"])
    ap.add_argument("--force_dp", action="store_true",
                    help="world size 1 only: create the RCCL process group anyway and run the data-parallel path (flat-bucket "
                         "all-reduce, all-gather with --gather_negatives, graph seams) through it")
    ap.add_argument("--no_parity", action="store_true")
    ap.add_argument("--no_alt_precision", action="store_true")
    ap.add_argument("--gather_negatives", action="store_true", help="BASELINE config 5: all-gather contrastive negatives")
    ap.add_argument("--graph", type=int, default=-1, help="replay the iteration as hipGraphs (default: on, for any number of ranks; 0 = eager launches)")
    ap.add_argument("--gen", type=str, default="", help="variant: override GEN.ENCODER_NAME (e.g. CONCEPT_OUTATTN_GEN)")
    ap.add_argument("--spec_norm", action="store_true", help="variant: DISC.SPEC_NORM=True (off in the headline cfg)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--no_entrypoint", action="store_true")
    a = ap.parse_args()
    w = WORKLOADS[a.workload]
    a.imsize = w["imsize"] if a.imsize is None else a.imsize
    a.batch = w["batch"] if a.batch is None else a.batch
    a.cfg = w["cfg"] if a.cfg is None else a.cfg
    return a


def pmc_traffic_lookup(imsize, batch, cfg_name, precision):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC summary of THIS workload
    (profiles/r03_pmc_hbm_traffic_<S>px_b<B>.csv: separate --pmc FETCH_SIZE / WRITE_SIZE passes of this script, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950; made by profiles/summarize.py).  Counters cannot be read from
    inside the process, so the figure is None for workloads without a committed summary."""
    import csv
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r{r:02d}_pmc_hbm_traffic_{imsize}px_b{batch}.csv") for r in (5, 4, 3))
                 if os.path.exists(q)), None)
    if cfg_name != "df_gan_damsm_nomagp.yml" or precision != "bf16" or path is None:
        return None
    rows = list(csv.DictReader(open(path)))
    # the summary was collected on an earlier build: say so when it names kernels the library no longer contains
    import re
    from xmc_gan_amd import lib as L
    try:
        blob = open(L.LIB_PATH, "rb").read()
        names = {m.group(1) for r in rows if "at::" not in r["kernel"] for m in [re.search(r"(?:\(anonymous namespace\)::)?([A-Za-z_][A-Za-z0-9_]*_kernel)\b", r["kernel"])] if m}
        gone = sorted(n for n in names if n.encode() not in blob)
        if gone:
            print(f"[bench] {os.path.basename(path)} names kernels that are not in {os.path.basename(L.LIB_PATH)} any more: {gone}; "
                  "`roofline.traffic` may be stale (re-collect with profiles/collect_r05.sh)", file=sys.stderr)
    except OSError:
        pass

    def lookup(kernel):
        # the library reports a kernel without trailing default template arguments; instantiations that differ only in the
        # epilogue-variant argument (EPI, common.h) share the reported name: those are averaged, weighted by their dispatches
        def avg(pred):
            tot, n = 0.0, 0
            for r in rows:
                if pred(r["kernel"]):
                    k = int(r["dispatches"])
                    tot += k * (float(r["fetch_MB_per_dispatch_corrected_x2"]) + float(r["write_MB_per_dispatch"]))
                    n += k
            return round(tot / n * 2**20) if n else None
        if not kernel.endswith(">"):                      # not a template: the name is followed by its argument list
            return avg(lambda nm: kernel + "(" in nm)
        exact = avg(lambda nm: any(kernel[:-1] + tail + "(" in nm for tail in (">", ", false>", ", false, false>")))
        if exact is not None:
            return exact
        # the same instantiation with further (defaulted / epilogue-variant) arguments: "<4, 4" must not match "<4, 40"
        return avg(lambda nm: kernel[:-1] + ", " in nm)
    return lookup


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota when there is one
    (os.cpu_count() reports the whole host and badly oversubscribes a container with a CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(cfg, cfg_name, imsize, seconds_budget=15.0):
    """Time the CPU oracle (oracle/xmc_ref.py, the parity checker) on this host: same cfg and image size, batch 8,
    as many whole iterations as fit in ~15 s (at least 2)."""
    import xmc_ref as X
    h = X.Hyper.from_cfg(cfg)
    cores = host_cores()
    torch.set_num_threads(cores)
    B = 8
    PG, PD = X.synth_params(X.gen_shapes(h), 1), X.synth_params(X.netd_shapes(h), 2)
    optG, optD = X.AdamState(h.g_lr, h.g_betas), X.AdamState(h.d_lr, h.d_betas)
    batch = X.synth_batch(h, B, seed=1)
    X.train_step(PG, PD, optG, optD, h, batch)          # warm-up (allocator, thread pool)
    t0, n = time.perf_counter(), 0
    while True:
        X.train_step(PG, PD, optG, optD, h, batch)
        n += 1
        dt = time.perf_counter() - t0
        if n >= 2 and (dt > seconds_budget or n >= 40):
            break
    return dict(value=round(B * n / dt, 4), unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n} full G+D iterations of oracle/xmc_ref.py (PyTorch CPU fp32), {imsize}x{imsize}, batch {B}, {cfg_name}")


def parity_leg(precision, cfg_name, size=64):
    """One G+D iteration at size x size, batch 8, NCH=32 in `precision` through the HIP kernels against the f32 CPU oracle on
    identical inputs and parameters (the reference's initialisation, block gammas 0.1 as in the timed run).  The oracle is the
    checker here, never the thing measured.  (256x256: ~4 s of oracle time on the GPU box's 16 host cores.)"""
    import xmc_ref as X
    from xmc_gan.config import gan
    import xmc_gan.train_gan as tg
    from xmc_gan_amd import ops
    from xmc_gan_amd.optim import HipAdam
    ops.set_precision(precision)
    gan.reset_cfg()
    gan.cfg_from_file(os.path.join(ROOT, "xmc_gan", "cfg", cfg_name))
    cfg = gan.cfg
    cfg.IMG.SIZE, cfg.TRAIN.BATCH_SIZE = size, 8
    h = X.Hyper.from_cfg(cfg)
    PG, PD = X.ref_init_params(X.gen_shapes(h), 1, 0.1), X.ref_init_params(X.netd_shapes(h), 2, 0.1)
    b = X.synth_batch(h, 8, seed=300, words_len=cfg.TEXT.MAX_LENGTH)
    torch.set_num_threads(host_cores())
    oG, oD = X.AdamState(h.g_lr, h.g_betas), X.AdamState(h.d_lr, h.d_betas)
    ref = X.train_step({k: v.clone() for k, v in PG.items()}, {k: v.clone() for k, v in PD.items()}, oG, oD, h, b)
    dev = torch.device("cuda", torch.cuda.current_device())
    netG = tg._GEN_ARCH[h.gen](cfg).to(dev)
    netD = tg._DISC_ARCH["DF_DISC"](cfg, is_disc=True).to(dev)
    netG.load_state_dict(PG, strict=True)
    netD.load_state_dict(PD, strict=True)
    with torch.no_grad():
        ps = netG.proj_sent(b["sent_embs"].to(dev))
        lr = netD.COND_DNET(netD(b["imgs"].to(dev)), sent_embs=ps)[0].float().cpu()
        lf = netD.COND_DNET(netD(ref["fake"].to(dev)), sent_embs=ps)[0].float().cpu()
    optG, optD = HipAdam(netG.parameters(), lr=h.g_lr, betas=h.g_betas), HipAdam(netD.parameters(), lr=h.d_lr, betas=h.d_betas)
    o = tg.gan_iteration(netG, netD, optG, optD, *(b[k].to(dev) for k in ("imgs", "sent_embs", "words_embs", "mask", "noise")), {})
    torch.cuda.synchronize()
    losses = {k: abs(float(o[k]) - float(ref[k])) / max(abs(float(ref[k])), 1e-12) for k in o
              if k != "fake" and k in ref and not torch.is_tensor(ref[k])}
    rel = lambda x, y: float((x - y).norm() / y.norm())
    worst = max(losses, key=losses.get)
    lg = {"real": round(rel(lr, ref["logit_real"]), 6), "fake_on_oracle_image": round(rel(lf, ref["logit_fake"]), 6)}
    del netG, netD, optG, optD
    return dict(mode={"bf16": "bf16", "f16": "f16", "fp32": "f32"}[precision], against="f32 CPU oracle (oracle/xmc_ref.py), same inputs and parameters",
                workload=f"one G+D iteration, {size}x{size}, batch 8, NCH=32, {cfg_name}, reference initialisation with block gammas 0.1",
                loss_rel_vs_f32_oracle=round(losses[worst], 6), worst_loss=worst,
                losses={k: round(v, 6) for k, v in losses.items()}, logit_rel=lg,
                bar=1e-3, losses_within_bar=bool(losses[worst] <= 1e-3), logits_within_bar=bool(max(lg.values()) <= 1e-3),
                within_bar=bool(losses[worst] <= 1e-3 and max(lg.values()) <= 1e-3))


def parity_legs(precision, cfg_name, size):
    """the leg at the benched image size, with the 64x64 one (the figure of rounds 1-3) beside it"""
    out = parity_leg(precision, cfg_name, size)
    if size != 64:
        small = parity_leg(precision, cfg_name, 64)
        out["at_64px"] = {k: small[k] for k in ("loss_rel_vs_f32_oracle", "worst_loss", "logit_rel", "within_bar")}
    return out


def alt_precision_leg(a):
    """The timed workload once more in the IEEE-half mode (a child process: a fresh allocator, graph pool and library state)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(a.steps), "--warmup", str(a.warmup),
           "--imsize", str(a.imsize), "--batch", str(a.batch), "--cfg", a.cfg, "--precision", "f16", "--no_cpu_baseline",
           "--no_parity", "--no_alt_precision", "--no_entrypoint"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if r.returncode != 0 or not line:
        return dict(dtype="f16", error=(r.stderr or r.stdout)[-400:])
    j = json.loads(line[-1])
    roof = j.get("roofline") or {}
    return dict(dtype="f16", value=j["value"], unit=j["unit"], ms_per_step=j["ms_per_step"], step_ms=j.get("step_ms"),
                step_frac_of_bf16_peak=j["step_frac_of_bf16_peak"], losses_finite=j["config"]["losses_finite"],
                loss_scale=j["config"].get("loss_scale"),
                roofline={k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launches", "avg_launch_us",
                                                   "traffic")} if roof else None,
                note="same kernels compiled for IEEE half (libxmc_gan_hip_f16.so): f16 MFMA = the bf16 rate; dynamic loss scale "
                     "with a found-inf skip inside the Adam kernel (initial 4096)")


def entrypoint_leg(a):
    """`python xmc_gan/train_gan.py --cfg <cfg> --synthetic N --bs B --imsize S --max_epoch 1` in a child process; train()'s own
    throughput report (from its fifth iteration to the end of the epoch, device-synchronised at both ends)."""
    import subprocess
    import tempfile
    n = 5 + max(a.steps, 20)
    with tempfile.TemporaryDirectory() as tmp:
        code = ("import json, sys; sys.path.insert(0, %r); import xmc_gan.train_gan as tg; "
                "last = tg.main(%r); print('ENTRYPOINT ' + json.dumps(last.get('throughput')))") % (
            ROOT, ["--cfg", os.path.join(ROOT, "xmc_gan", "cfg", a.cfg), "--synthetic", str(n), "--bs", str(a.batch), "--imsize", str(a.imsize),
                   "--max_epoch", "1", "--precision", a.precision, "--output_dir", os.path.join(tmp, "run")])
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("ENTRYPOINT ")]
    if r.returncode != 0 or not line:
        return dict(error=(r.stderr or r.stdout)[-400:])
    thr = json.loads(line[-1][len("ENTRYPOINT "):]) or {}
    return dict(command=f"python xmc_gan/train_gan.py --cfg xmc_gan/cfg/{a.cfg} --synthetic {n} --bs {a.batch} --imsize {a.imsize} --max_epoch 1 "
                        f"--precision {a.precision}", images_per_s=thr.get("images_per_s"), ms_per_step=thr.get("ms_per_step"),
                steps=thr.get("steps"), hipgraph=thr.get("hipgraph"),
                note="loader batches are host tensors (pinned, uploaded on a side stream while the previous iteration runs); includes the "
                     "text encoder's forward and the CPU-side noise draw of every iteration (train_gan.py:160-170,197)")


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD `torch.distributed.run` (this process has
    not initialised the GPU and never does), relay the ranks' output, pass rank 0's JSON line through, and fail unless the
    line reports n_gpus == N."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc != 0:
        raise SystemExit(rc)
    if line is None or json.loads(line).get("n_gpus") != n:
        raise SystemExit(f"bench.py --gpus {n}: the launched ranks did not report n_gpus == {n}")
    print(line, flush=True)
    return 0


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a.gpus)                       # before anything touches the GPU in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: start {a.gpus} ranks (python bench.py --gpus {a.gpus} "
                         "launches them itself) or pass --gpus equal to the number of ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path (the CPU oracle is only the baseline leg)")
    backend = os.environ.get("XMC_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        if backend == "nccl":
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible; RCCL needs one GPU per rank "
                             "(XMC_DIST_BACKEND=gloo lets rehearsal ranks share a card)")
        local_rank %= max(ndev, 1)                       # gloo rehearsal: several ranks on one card
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or a.force_dp:
        if world == 1:                                   # single-card rehearsal of the RCCL path: a one-rank group of our own
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
        if a.force_dp:
            from xmc_gan_amd import parallel
            parallel.force_collectives(True)
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()

    from xmc_gan.config import gan
    import xmc_gan.train_gan as tg
    from xmc_gan_amd import ops, prof
    ops.set_precision(a.precision)
    gan.reset_cfg()
    gan.cfg_from_file(os.path.join(ROOT, "xmc_gan", "cfg", a.cfg))
    cfg = gan.cfg
    cfg.IMG.SIZE, cfg.TRAIN.BATCH_SIZE = a.imsize, a.batch
    if a.spec_norm:
        cfg.DISC.SPEC_NORM = True
    if a.gen:
        cfg.GEN.ENCODER_NAME = a.gen
    torch.manual_seed(100 + rank)
    netG, netD, optG, optD = tg.build_models(dev)
    if world > 1:
        for p_ in list(netG.parameters()) + list(netD.parameters()) + list(netG.buffers()) + list(netD.buffers()):
            torch.distributed.broadcast(p_.data, 0)
    # block gammas start at 0 in the reference (df_gan.py:195,281); give them a value so no branch is dead weight
    with torch.no_grad():
        for n_, p_ in list(netG.named_parameters()) + list(netD.named_parameters()):
            if n_.endswith("gamma"):
                p_.fill_(0.1)

    B, S, E, T = a.batch, a.imsize, cfg.TEXT.EMBEDDING_DIM, cfg.TEXT.MAX_LENGTH
    g = torch.Generator().manual_seed(100 + rank)
    nb = 4                                                    # a few distinct resident batches, cycled
    data = []
    for _ in range(nb):
        lens = torch.randint(5, T + 1, (B,), generator=g)
        data.append(dict(imgs=(torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev),
                         sent=torch.randn(B, E, generator=g).to(dev), words=torch.randn(B, E, T, generator=g).to(dev),
                         mask=(torch.arange(T)[None, :] >= lens[:, None]).to(dev),
                         noise=torch.randn(B, cfg.TRAIN.NOISE_DIM, generator=g).to(dev)))
    opts = tg.StepOptions(gather_negatives=a.gather_negatives)
    state = {}

    def eager(imgs, sent, words, mask, noise, st):
        return tg.gan_iteration(netG, netD, optG, optD, imgs, sent, words, mask, noise, st, opts)

    # hipGraph replay for any number of ranks: the collectives are eager seams between captured segments (xmc_gan_amd/graph.py)
    use_graph = True if a.graph < 0 else bool(a.graph)
    graphed = None
    if use_graph:
        from xmc_gan_amd.graph import GraphedIteration
        d0 = data[0]
        ok = 1
        try:
            graphed = GraphedIteration(eager, (d0["imgs"], d0["sent"], d0["words"], d0["mask"], d0["noise"]),
                                       n_critic=cfg.TRAIN.N_CRITIC, warmup=2)
            for i in range(3):                   # 2 eager warm-ups + the capture itself, outside the timed region
                graphed(d0["imgs"], d0["sent"], d0["words"], d0["mask"], d0["noise"])
            torch.cuda.synchronize()
        except Exception as e:                   # noqa: BLE001
            import traceback
            print(f"[bench rank {rank}] graph capture failed ({type(e).__name__}: {e})\n"
                  + "".join(traceback.format_exc().splitlines(True)[-12:]), file=sys.stderr)
            if dist_on:
                # A rank that threw mid-capture has issued fewer seam collectives than its peers; anything it sends now (a
                # "did it work" flag included) would be matched against a peer's pending gradient all-reduce of another size and
                # dtype -- undefined behaviour or a hang.  There is no safe in-band way to agree on a fallback: fail the job.
                os._exit(3)
            ok = 0
        if not ok:
            graphed, use_graph = None, False             # single process: packed-weight caches were invalidated by graph._capture

    def step(i, force_eager=False):
        d = data[i % nb]
        if graphed is not None and not force_eager:
            return graphed(d["imgs"], d["sent"], d["words"], d["mask"], d["noise"])
        return eager(d["imgs"], d["sent"], d["words"], d["mask"], d["noise"], state)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    # per-step HIP events ride along on the launch stream (recorded, never waited on inside the timed region): the spread of the
    # K steps.  `value` stays the contract's K steps / wall time between the two barriers.
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        last = step(a.warmup + i)
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    finite = all(torch.isfinite(v).all().item() for k, v in last.items() if k != "fake")

    roof = roof_hbm = None
    if not a.no_roofline:
        # one more iteration (on EVERY rank: it contains the collectives) with each conv launch bracketed by HIP
        # events on the launch stream; rank 0 reports
        prof.enable()
        step(a.warmup + a.steps, force_eager=True)
        torch.cuda.synchronize()
        peak = PEAK_F32_TFLOPS if a.precision == "fp32" else PEAK_BF16_TFLOPS      # f16 MFMA = the bf16 rate
        roof = prof.summary(peak, pmc_traffic_lookup(S, B, a.cfg, a.precision))
        roof_hbm = prof.summary_hbm(PEAK_HBM_GBS, 1e3 * peak / PEAK_HBM_GBS)
        if os.environ.get("XMC_PROF_SHAPES") and rank == 0:
            for fam, tag, n, ms, tf, tbs in prof.by_shape()[:max(60, int(os.environ["XMC_PROF_SHAPES"]))]:
                print(f"{fam:52s} {tag:58s} n={n:3d} {ms:8.3f} ms {tf:8.1f} TF/s {tbs:6.2f} TB/s", file=sys.stderr)
            print(f"max memory allocated: {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", file=sys.stderr)
        prof.disable()
        barrier()

    if rank == 0:
        magp = bool(cfg.TRAIN.MAGP)
        imgs_s = world * B * a.steps / dt
        gf = STEP_GFLOP.get((S, magp))
        if cfg.GEN.ENCODER_NAME == "CONCEPT_IN_DF_GEN" and not magp and S in G_IN_GFLOP:
            gf = round(9 * D_GFLOP[S] + 3 * G_IN_GFLOP[S], 2)       # CONCEPT_IN_DF_GEN replaces G's forward count
        elif cfg.GEN.ENCODER_NAME != "DF_GEN":
            gf = None
        out = dict(metric="images/sec per G+D step", value=round(imgs_s, 2), unit="images/s", n_gpus=world, steps=a.steps,
                   warmup=a.warmup, ms_per_step=round(1e3 * dt / a.steps, 3),
                   step_ms=dict(median=round(per_step[len(per_step) // 2], 3), min=round(per_step[0], 3), max=round(per_step[-1], 3),
                                images_per_s_at_median=round(world * B / per_step[len(per_step) // 2] * 1e3, 1)),
                   higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype={"bf16": "bf16", "f16": "f16", "fp32": "f32"}[a.precision], data="synthetic",
                   config=dict(workload=f"{S}x{S} COCO-shaped synthetic batch, {B} images per GPU, one full G+D iteration "
                                        f"(D step{' + MA-GP' if magp else ''} + G step + Adam x{3 if magp else 2}), {a.cfg}"
                                        + (" with DISC.SPEC_NORM=True" if a.spec_norm else "")
                                        + (f" with GEN.ENCODER_NAME={a.gen}" if a.gen else ""),
                               per_gpu_batch=B, global_batch=B * world, image_size=S, cfg=a.cfg,
                               parallelism=f"dp{world}" + ("+gather" if a.gather_negatives else ""), hipgraph=bool(use_graph),
                               losses_finite=finite, workload_name=a.workload,
                               loss_scale=(ops.loss_scaler_stats() or None)),
                   step_algorithmic_tflops=None if gf is None else round(imgs_s * gf / 1e3, 2),
                   step_frac_of_bf16_peak=None if gf is None else round(imgs_s * gf / 1e3 / (PEAK_BF16_TFLOPS * world), 4))
        if roof is not None:
            out["roofline"] = roof
            out["roofline_hbm"] = roof_hbm
        out["dist"] = dict(backend=(torch.distributed.get_backend() if dist_on else None), world_size=world,
                           process_group=bool(dist_on), forced_at_world_1=bool(a.force_dp),
                           rccl_version=(".".join(map(str, torch.cuda.nccl.version())) if dist_on and backend == "nccl" else None),
                           collectives_per_iteration=(graphed.seams_per_iteration() if graphed is not None else None),
                           # host time inside one collective call at replay (RCCL: the enqueue; gloo: the whole exchange), rank 0
                           seam_host_ms=(round(1e3 * graphed.seam_host_s[0] / graphed.seam_host_s[1], 4)
                                         if graphed is not None and graphed.seam_host_s[1] else None))
        if not a.no_cpu_baseline and world == 1:       # the CPU leg is timed on rank 0 of the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(cfg, a.cfg, S)
    # free the timed run's networks, graphs and pools before the side legs
    del graphed, netG, netD, optG, optD, data, last
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not a.force_dp:
        if not a.no_alt_precision and a.precision == "bf16" and not a.gen and not a.spec_norm:
            out["alt_precision"] = alt_precision_leg(a)
        if not a.no_entrypoint and not a.gen and not a.spec_norm:
            out["entrypoint"] = entrypoint_leg(a)
            if out["entrypoint"].get("images_per_s"):
                out["entrypoint"]["frac_of_value"] = round(out["entrypoint"]["images_per_s"] / out["value"], 4)
        if not a.no_parity and not a.gen and not a.spec_norm:
            out["parity"] = parity_legs(a.precision, a.cfg, a.imsize)
            if "alt_precision" in out:
                out["alt_precision"]["parity"] = parity_legs("f16", a.cfg, a.imsize)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
