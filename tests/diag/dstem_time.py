"""Timing aid (GPU): the composed discriminator stem (csrc/dstem.hip) against the kernels it replaces, at the bench's shapes.
usage: python tests/diag/dstem_time.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xmc_gan_amd import ops, lib as L

dev, dt = torch.device("cuda"), torch.bfloat16
ops.set_precision("bf16")


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
    print(f"   {name:52s} {e0.elapsed_time(e1) / n:8.3f} ms  [{L.load().xmc_last_kernel().decode()}]")


for N in (256, 512):
    H = W = 256
    x = (torch.rand(N, H, W, 8, device=dev) * 2 - 1).to(dt)
    x[..., 3:] = 0
    w_img, b_img = torch.randn(32, 3, 3, 3, device=dev) * 0.2, torch.randn(32, device=dev) * 0.1
    w0 = torch.randn(64, 32, 4, 4, device=dev) * 0.05
    ws, bs = torch.randn(64, 32, 1, 1, device=dev) * 0.2, torch.randn(64, device=dev) * 0.1
    g_img, g0, gs = ops.ConvGeom(3, 32, 3, 1, 1), ops.ConvGeom(32, 64, 4, 2, 1), ops.ConvGeom(32, 64, 1, 1, 0)
    wsets, bias, D, DB = ops._dstem_compose_raw(w_img, b_img, w0, ws, bs)
    timeit("composition of the weights (4 tables)", lambda: ops._dstem_compose_raw(w_img, b_img, w0, ws, bs))
    bi, bsp = ops._bias_padded(b_img, g_img), ops._bias_padded(bs, gs)
    print(f"N{N} {H}x{W}")
    timeit("composed stem forward (h1, sc)", lambda: ops._dstem_fwd_raw(x, wsets, bias))
    timeit("  ... h1 only (shortcut recomputed by the block end)", lambda: ops._dstem_fwd_raw(x, wsets, bias, want_sc=False))
    h1_, _ = ops._dstem_fwd_raw(x, wsets, bias)
    timeit("  + border pixels of h1", lambda: ops._dstem_border_fwd_raw(x, wsets, bias, D, DB, h1_))
    ci, cip = ops._conv_fwd_raw(x, w_img, bi, g_img, L.ACT_NONE, dt, want_pool=True)
    timeit("  replaces: conv_img (+ pooled output)", lambda: ops._conv_fwd_raw(x, w_img, bi, g_img, L.ACT_NONE, dt, want_pool=True))
    timeit("            conv_r[0] 4x4 s2 32 -> 64", lambda: ops._conv_fwd_raw(ci, w0, None, g0, L.ACT_LRELU, dt))
    timeit("            conv_s 1x1 on the pooled map", lambda: ops._conv_fwd_raw(cip, ws, bsp, gs, L.ACT_NONE, dt))
    dh1 = torch.randn(N, H // 2, W // 2, 64, device=dev).to(dt)
    dsc = torch.randn(N, H // 2, W // 2, 64, device=dev).to(dt)

    def wg():
        ops.new_iteration(dev)
        return ops._dstem_wgrad_raw(x, dh1, dsc)
    timeit("composed stem weight gradient (+ border tables)", wg)
    tabs = wg()
    timeit("  + adjoint of the composition", lambda: ops._dstem_compose_bwd_raw(w_img, b_img, w0, ws, bs, *tabs))

    def old_w0():
        ops.new_iteration(dev)
        return ops._conv_wgrad_raw(ci, dh1, g0)
    timeit("  replaces: wgrad conv_r[0]", old_w0)
    dci = torch.randn_like(ci)

    def old_wi():
        ops.new_iteration(dev)
        return ops._conv_wgrad_raw(x, dci, g_img, want_bias=True)
    timeit("            wgrad conv_img", old_wi)
    timeit("            dgrad conv_r[0] (d conv_img out)", lambda: ops._conv_dgrad_raw(dh1, w0, g0, (H, W), dt))

    timeit("composed stem image gradient (+ border)", lambda: ops._dstem_dgrad_raw(dh1, dsc, wsets, D, H, W))
    timeit("  replaces: dgrad conv_img (d image)", lambda: ops._conv_dgrad_raw(dci, w_img, g_img, (H, W), dt))
    timeit("            dgrad conv_s (d pooled map)", lambda: ops._conv_dgrad_raw(dsc, ws, gs, (H // 2, W // 2), dt))

    def old_ws():
        ops.new_iteration(dev)
        return ops._conv_wgrad_raw(cip, dsc, gs, want_bias=True)
    timeit("            wgrad conv_s", old_ws)
    del x, ci, cip, dh1, dsc, dci
    torch.cuda.empty_cache()
