"""Timing aid (GPU): one convolution shape with the epilogue options switched on one at a time.
usage: python tests/diag/epilogue_cost.py [N H C]"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xmc_gan_amd import ops, lib as L

N, H, Cc = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 256, 32)
dev = torch.device("cuda")
dt = torch.bfloat16
x = torch.randn(N, H, H, Cc, device=dev).to(dt)
w = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05
b = torch.zeros(Cc, device=dev)
geom = ops.ConvGeom(Cc, Cc, 3, 1, 1)
sc = torch.randn(N, H // 2, H // 2, Cc, device=dev).to(dt)
scf = torch.randn(N, H, H, Cc, device=dev).to(dt)
al = torch.full((1,), 0.5, device=dev)


def timeit(name, fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:58s} {e0.elapsed_time(e1) / n:8.3f} ms   {L.load().xmc_last_kernel().decode()}")


R = ops._conv_fwd_raw
timeit("plain", lambda: R(x, w, b, geom, L.ACT_NONE, dt))
timeit("lrelu", lambda: R(x, w, b, geom, L.ACT_LRELU, dt))
timeit("res (same layout)", lambda: R(x, w, b, geom, L.ACT_NONE, dt, res=scf))
timeit("res half-res (res_mode 2)", lambda: R(x, w, b, geom, L.ACT_NONE, dt, res=sc, res_mode=2))
timeit("res_mode 2 + alpha", lambda: R(x, w, b, geom, L.ACT_NONE, dt, res=sc, res_mode=2, alpha=al))
timeit("res_mode 2 + alpha + round_act", lambda: R(x, w, b, geom, L.ACT_NONE, dt, res=sc, res_mode=2, alpha=al, round_act=True))
timeit("res_mode 2 + alpha + round_act + post_act", lambda: R(x, w, b, geom, L.ACT_NONE, dt, res=sc, res_mode=2, alpha=al, round_act=True, post_act=L.ACT_LRELU))
timeit("alpha only", lambda: R(x, w, b, geom, L.ACT_NONE, dt, alpha=al))
timeit("want2", lambda: R(x, w, b, geom, L.ACT_NONE, dt, want2=True))
timeit("mask (dgrad-like)", lambda: R(x, w, b, geom, L.ACT_NONE, dt, mask=scf))

# the same fused call in its neighbourhood (affine pair in front, output convolution behind), timed alone inside the sequence
g0 = torch.ones(N, Cc, device=dev); b0 = torch.zeros(N, Cc, device=dev)
wo = torch.randn(3, Cc, 3, 3, device=dev) * 0.05
geo = ops.ConvGeom(Cc, 3, 3, 1, 1)
bo = torch.zeros(8, device=dev)
for rep in range(3):
    h2 = ops._affine_fwd_raw(x, [g0, b0, g0, b0], 0.2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = R(h2, w, b, geom, L.ACT_NONE, dt, res=sc, res_mode=2, alpha=al, round_act=True, post_act=L.ACT_LRELU)
    e1.record()
    img = R(y, wo, bo, geo, L.ACT_TANH, dt)
    torch.cuda.synchronize()
    print(f"in sequence: fused c2 {e0.elapsed_time(e1):.3f} ms")
al0 = torch.full((1,), 0.1, device=dev)
timeit("fused, alpha 0.1, input = affine output", lambda: R(h2, w, b, geom, L.ACT_NONE, dt, res=sc, res_mode=2, alpha=al0, round_act=True, post_act=L.ACT_LRELU))
