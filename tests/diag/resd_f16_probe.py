import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from xmc_gan_amd import ops
from xmc_gan.model.df_gan import resD
DEV = "cuda"
mode = sys.argv[1] if len(sys.argv) > 1 else "f16"
ops.set_precision(mode)
dt = ops.act_dtype()
rel = lambda a, b: ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()
for (cin, cout, H, gscale) in [(8, 16, 64, 1.0), (16, 32, 32, 1.0), (32, 64, 16, 1.0), (64, 128, 8, 1.0), (8, 16, 64, 4096.0), (32, 64, 16, 4096.0)]:
    torch.manual_seed(cin + H)
    blk = resD(cin, cout, downsample=True).to(DEV)
    with torch.no_grad():
        blk.gamma.fill_(0.1)
    x0 = torch.randn(4, H, H, ops.chan_pad(cin, dt), device=DEV).to(dt)
    r = (torch.randn(4, H // 2, H // 2, ops.pad_to(cout, 8), device=DEV) * gscale).to(dt)
    got = {}
    for name in ("bits", "values", "composed"):
        blk.zero_grad()
        x = x0.clone().requires_grad_()
        with (ops.composable() if name == "composed" else ops.second_order(name == "values")):
            y = blk(x)
        y.backward(r)
        got[name] = [y.detach(), x.grad] + [p.grad.clone() for p in blk.parameters() if p.grad is not None]
    names = ["y", "dx"] + [n for n, p in blk.named_parameters() if p.grad is not None]
    print(f"{mode} {cin}->{cout} H{H} gscale {gscale}")
    for n, a, b, c in zip(names, got["bits"], got["values"], got["composed"]):
        print(f"    {n:16s} bits vs composed {rel(a, c):.2e}   values vs composed {rel(b, c):.2e}")
