"""Timing aid (GPU): the stem block's masked gradient, written by xmc_signmask_apply and read twice vs applied while its two
consumers stage dout (XmcConvDesc.mask_bits).  N = 256, 128x128, 64 -> 64.
usage: python tests/diag/stage_bits_time.py"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xmc_gan_amd import ops, lib as L

dev = torch.device("cuda")
ops.set_precision("bf16")
dt = ops.act_dtype()
N, H, C = 256, 128, 64
dout = torch.randn(N, H, H, C, device=dev).to(dt)
h1 = torch.randn(N, H, H, C, device=dev).to(dt)
bits = torch.randint(0, 256, (N, H, H, C // 8), device=dev, dtype=torch.int64).to(torch.uint8)
w2 = torch.randn(C, C, 3, 3, device=dev) * 0.03
g2 = ops.ConvGeom(C, C, 3, 1, 1)
al = torch.full((1,), 0.5, device=dev)
dot = torch.zeros(1, device=dev)
gr = torch.empty_like(dout)


def timeit(name, fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"   {name:44s} {e0.elapsed_time(e1) / n * 1000:8.1f} us  {L.load().xmc_last_kernel().decode()}")


def mask():
    L.call("xmc_signmask_apply", dout.data_ptr(), bits.data_ptr(), gr.data_ptr(), dout.numel(), 0.2, ops._code(dt), ops._st())


timeit("mask pass", mask)
timeit("data gradient (mask, alpha, dot) of gr", lambda: ops._conv_dgrad_raw(gr, w2, g2, (H, H), dt, mask=h1, alpha=al, dot=dot))
timeit("  ... of dout with the bits staged", lambda: ops._conv_dgrad_raw(ops._StagedMask(dout, bits), w2, g2, (H, H), dt, mask=h1, alpha=al, dot=dot))
timeit("weight gradient of gr", lambda: (ops._arena.new_iteration(dev), ops._conv_wgrad_raw(h1, gr, g2, scale=al)))
timeit("  ... of dout with the bits staged", lambda: (ops._arena.new_iteration(dev), ops._conv_wgrad_raw(h1, ops._StagedMask(dout, bits), g2, scale=al)))
