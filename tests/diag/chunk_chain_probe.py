"""Does the 256 MB memory-side cache reward running a producer -> consumer chain over batch CHUNKS?  The stem block's forward chain
(stem forward -> border pixels -> conv_r[2] block end with the recomputed shortcut) over 512 images in chunks of 512 / 256 / 128 / 64 / 32:
total time of the chain.  usage: python tests/diag/chunk_chain_probe.py"""
import sys, os, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xmc_gan_amd import ops, lib as L

dev = torch.device("cuda")
ops.set_precision("bf16")
dt = ops.act_dtype()
N, H = 512, 256
g = torch.Generator().manual_seed(0)
xin = torch.zeros(N, H, H, 8, dtype=dt, device=dev)
xin[..., :3] = (torch.rand(N, H, H, 3, generator=g) * 2 - 1).to(dt).to(dev)
w_img, b_img = (torch.randn(32, 3, 3, 3, generator=g) / math.sqrt(27)).to(dev), (torch.randn(32, generator=g) * 0.1).to(dev)
w0 = (torch.randn(64, 32, 4, 4, generator=g) / math.sqrt(512)).to(dev)
ws, bs = (torch.randn(64, 32, 1, 1, generator=g) / math.sqrt(32)).to(dev), (torch.randn(64, generator=g) * 0.1).to(dev)
w2 = (torch.randn(64, 64, 3, 3, generator=g) / math.sqrt(576)).to(dev)
wsets, bias, D, DB = ops.compose_dstem(w_img, b_img, w0, ws, bs)
g2 = ops.ConvGeom(64, 64, 3, 1, 1)
al = torch.tensor([0.5], device=dev)
out = torch.empty(N, H // 2, H // 2, 64, dtype=dt, device=dev)


def chain(chunk):
    for n0 in range(0, N, chunk):
        x = xin[n0:n0 + chunk]
        h1, _ = ops._dstem_fwd_raw(x, wsets, bias, want_sc=False)
        ops._dstem_border_fwd_raw(x, wsets, bias, D, DB, h1)
        ops._conv_fwd_raw(h1, w2, None, g2, L.ACT_LRELU, dt, alpha=al, want_sign=True, want_pool=True, round_act=True,
                          sc_img=ops._dstem_sc_operands(x, wsets, bias), out=out[n0:n0 + chunk])


for chunk in (512, 256, 128, 64, 32):
    for _ in range(2):
        chain(chunk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        chain(chunk)
    e1.record()
    torch.cuda.synchronize()
    print(f"chunk {chunk:4d}: {e0.elapsed_time(e1) / 5:7.3f} ms for the chain over {N} images")
