"""diagnostic: where does a convolution case differ from F.conv2d?  python tests/diag/w3_debug.py mode cin cout k s p H N [dgrad]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from xmc_gan_amd import ops
from xmc_gan_amd import lib as L
mode = sys.argv[1]
cin, cout, k, s, p, H, N = [int(v) for v in sys.argv[2:9]]
ops.set_precision(mode)
dt = ops.act_dtype()
g = torch.Generator().manual_seed(1)
x = torch.randn(N, cin, H, H, generator=g).to(dt).float()
w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).to(dt).float()
geom = ops.ConvGeom(cin, cout, k, s, p)
OH = geom.out_hw(H, H)[0]
r = torch.randn(N, cout, OH, OH, generator=g).to(dt).float()
xr = x.clone().requires_grad_()
yr = F.conv2d(xr, w, None, s, p)
(yr * r).sum().backward()
xd = x.permute(0, 2, 3, 1).contiguous().to("cuda", dt)
wd = torch.nn.Parameter(w.cuda())
for it in range(3):
    y = ops._conv_fwd_raw(xd, wd, None, geom, 0, dt)
    kf = L.load().xmc_last_kernel().decode()
    dx = ops._conv_dgrad_raw(r.permute(0, 2, 3, 1).contiguous().to("cuda", dt), wd, geom, (H, H), dt)
    kd = L.load().xmc_last_kernel().decode()
    for name, got, ref, kern in (("fwd", y, yr.detach(), kf), ("dgrad", dx, xr.grad, kd)):
        got = got.float().cpu().permute(0, 3, 1, 2)
        err = (got - ref).abs()
        tol = 4e-3 * ref.abs().max() if mode == "f16" else 3e-2 * ref.abs().max()
        bad = (err > tol).nonzero()
        print(f"[{mode} it{it}] {name} {kern}: max err {err.max():.4f} (scale {ref.abs().max():.3f}), bad {len(bad)}")
        if len(bad):
            n_, c_, y_, x_ = bad.unbind(1)
            print("   samples", sorted(set(n_.tolist())), "rows", sorted(set(y_.tolist())), "cols", sorted(set(x_.tolist()))[:40], "chans", sorted(set(c_.tolist()))[:70])
