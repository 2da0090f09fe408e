"""Outputs of the 128 -> 64 channel class data gradient / fused upsample convolution under the current dispatch, saved to /tmp/s128/<tag>.pt on
the GPU box: run once as is and once with XMC_DEBUG_DISPATCH=no_ptile_slab128, then tests/diag/s128_swap_compare.py (round 4: the two
kernels differ by one ulp on 0.09 % of the elements; tests/diag/magp_swap_sensitivity.sh shows what that does to the MA-GP G-step loss).
usage: python tests/diag/s128_swap_outputs.py <tag>"""
import sys, os, torch, math
sys.path.insert(0, "/root/repo")
import torch.nn.functional as F
from xmc_gan_amd import ops, lib as L
dev = torch.device("cuda")
tag = sys.argv[1]
out = {}
for mode in ("f16",):
    ops.set_precision(mode); dt = ops.act_dtype()
    g = torch.Generator().manual_seed(1)
    for (N, H) in ((8, 128), (2, 256)):
        cin, cout = 64, 128
        OH = H // 2
        gh = torch.randn(N, OH, OH, cout, generator=g).to(dt).to(dev)
        dxp = torch.randn(N, OH, OH, cin, generator=g).to(dt).to(dev)
        mask = torch.randn(N, H, H, cin, generator=g).to(dt).to(dev)
        w0 = (torch.randn(cout, cin, 4, 4, generator=g) / math.sqrt(cin * 16)).to(dev)
        al = torch.tensor([0.37], device=dev)
        g0 = ops.ConvGeom(cin, cout, 4, 2, 1)
        for name, kw in (("plain", {}), ("res rows", dict(res=dxp, res_rows=True, res_scale=0.25)), ("mask", dict(mask=mask)), ("mask+alpha", dict(mask=mask, alpha=al)),
                         ("res rows+mask", dict(res=dxp, res_rows=True, res_scale=0.25, mask=mask))):
            got = ops._conv_dgrad_raw(gh, w0, g0, (H, H), dt, **kw)
            out[f"{N}_{H}_{name}"] = got.float().cpu()
            print(tag, N, H, name, L.load().xmc_last_kernel().decode())
        # the fused upsample convolution forward (128 -> 64)
        x = torch.randn(N, OH, OH, 128, generator=g).to(dt).to(dev)
        w = (torch.randn(64, 128, 3, 3, generator=g) / math.sqrt(128 * 9)).to(dev)
        b = (torch.randn(64, generator=g) * 0.1).to(dev)
        y = ops.upconv3x3(x, w, b, ops.ConvGeom(128, 64, 3, 1, 1))
        out[f"{N}_{H}_upconv"] = y.float().cpu()
        print(tag, "upconv", L.load().xmc_last_kernel().decode())
os.makedirs("/tmp/s128", exist_ok=True)
torch.save(out, f"/tmp/s128/{tag}.pt")
