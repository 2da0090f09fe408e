"""Timing aid (GPU): the discriminator block-end convolution with its compile-time epilogue sets, per kernel family.
usage: python tests/diag/blockend_cost.py"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xmc_gan_amd import ops, lib as L

dev = torch.device("cuda")
dt = torch.bfloat16
ops.set_precision("bf16")


def timeit(name, fn, gflop, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"   {name:44s} {ms:8.3f} ms {gflop / ms:8.1f} TF/s  {L.load().xmc_last_kernel().decode()}")


R = ops._conv_fwd_raw
for (N, H, C) in [(512, 128, 64), (256, 128, 64), (512, 64, 128), (256, 64, 128), (512, 32, 256), (256, 32, 256), (512, 16, 512)]:
    x = torch.randn(N, H, H, C, device=dev).to(dt)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.03
    geom = ops.ConvGeom(C, C, 3, 1, 1)
    sc = torch.randn(N, H, H, C, device=dev).to(dt)
    al = torch.full((1,), 0.5, device=dev)
    dgam = torch.zeros(1, device=dev)
    gf = 2.0 * N * H * H * C * C * 9 / 1e9
    print(f"N{N} {H}x{H} {C}->{C}")
    timeit("plain", lambda: R(x, w, None, geom, L.ACT_NONE, dt), gf)
    timeit("mask (data gradient through a LeakyReLU)", lambda: R(x, w, None, geom, L.ACT_NONE, dt, mask=sc), gf)
    timeit("mask + alpha + dot (DgDot)", lambda: ops._conv_dgrad_raw(x, w, geom, (H, H), dt, mask=sc, alpha=al, dot=dgam), gf)
    timeit("lrelu+round+alpha+res (no set: generic)", lambda: R(x, w, None, geom, L.ACT_LRELU, dt, res=sc, alpha=al, round_act=True), gf)
    timeit("DLastS: + sign", lambda: R(x, w, None, geom, L.ACT_LRELU, dt, res=sc, alpha=al, round_act=True, want_sign=True), gf)
    timeit("DFwd:   + pool", lambda: R(x, w, None, geom, L.ACT_LRELU, dt, res=sc, alpha=al, round_act=True, want_pool=True), gf)
    timeit("DKeepS: + pool + sign", lambda: R(x, w, None, geom, L.ACT_LRELU, dt, res=sc, alpha=al, round_act=True, want_pool=True, want_sign=True), gf)
    timeit("DKeep:  + pool + dst2", lambda: R(x, w, None, geom, L.ACT_LRELU, dt, res=sc, alpha=al, round_act=True, want_pool=True, want2=True), gf)
    timeit("DLast:  + dst2", lambda: R(x, w, None, geom, L.ACT_LRELU, dt, res=sc, alpha=al, round_act=True, want2=True), gf)
    del x, sc
    torch.cuda.empty_cache()
