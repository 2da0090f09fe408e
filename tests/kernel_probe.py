"""Micro-driver: times convolution shapes in isolation through the C ABI (not a test).

    python tests/kernel_probe.py "N,H,cin,cout,k,s,p,mode[,reps]" ...      mode: fwd | dgrad | wgrad | upfwd | updgrad | upwgrad
    XMC_LIB_PATH=/path/to/variant.so python tests/kernel_probe.py ...     (A/B of kernel variants: one process per library)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xmc_gan_amd import ops
from xmc_gan_amd import lib as L


def run(spec):
    f = spec.split(",")
    N, H, cin, cout, k, s, p = [int(v) for v in f[:7]]
    mode = f[7] if len(f) > 7 else "fwd"
    reps = int(f[8]) if len(f) > 8 else 20
    g = torch.Generator().manual_seed(0)
    bf = torch.bfloat16
    geom = ops.ConvGeom(cin, cout, k, s, p)
    x = torch.randn(N, H, H, ops.chan_pad(cin, bf), generator=g).to("cuda", bf)
    w = torch.nn.Parameter((torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda())
    if mode in ("upfwd", "updgrad", "upwgrad"):
        OH = 2 * H
    else:
        OH = geom.out_hw(H, H)[0]
    dy = torch.randn(N, OH, OH, ops.pad_to(cout, 8), generator=g).to("cuda", bf)

    clk = os.environ.get("XMC_PROBE_CLOCK")      # ablation libraries built with WT_ABL & 32 write (cycles, 100 MHz ticks) per
    dbg = torch.zeros(1 << 16, dtype=torch.float32, device="cuda") if clk else None   # workgroup into the bias pointer

    def once():
        if mode == "fwd":
            return ops._conv_fwd_raw(x, w, dbg, geom, 0, bf)
        if mode == "dgrad":
            return ops._conv_dgrad_raw(dy, w, geom, (H, H), bf)
        if mode == "wgrad":
            ops.new_iteration(x.device)
            return ops._conv_wgrad_raw(x, dy, geom)
        if mode == "upwgrad":
            ops.new_iteration(x.device)
            return ops._conv_wgrad_raw(x, dy, geom, up=True)
        if mode == "upfwd":
            return ops._upconv_fwd_raw(x, w, None, geom, 0, bf)
        if mode == "updgrad":
            return ops._upconv_dgrad_raw(dy, w, geom, bf)
        raise ValueError(mode)

    once(); once()
    kern = L.load().xmc_last_kernel().decode()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            once()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    fl = 2.0 * N * OH * OH * cin * cout * k * k
    extra = ""
    if clk:
        v = dbg.view(torch.int64)[:512].cpu().view(-1, 2)
        v = v[(v[:, 1] > 0)]
        if len(v):
            mhz = (v[:, 0].double() / v[:, 1].double() * 100.0)
            extra = f"  clock {mhz.median().item():.0f} MHz (min {mhz.min().item():.0f}, max {mhz.max().item():.0f}; {len(v)} wgs, {v[:,0].double().median().item():.0f} cycles)"
    print(f"{mode:7s} N{N} {H}x{H} {cin}->{cout} k{k}s{s}: {best:7.3f} ms {fl / best / 1e9:7.1f} TF/s  [{kern}]{extra}", flush=True)


if __name__ == "__main__":
    for spec in sys.argv[1:]:
        run(spec)
