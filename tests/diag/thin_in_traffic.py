"""conv_img forward (3 -> 32 @256^2, batch 256) with and without the pooled third output, once each, for a rocprofv3 --pmc pass:
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d OUT -- python3 tests/diag/thin_in_traffic.py"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xmc_gan_amd import ops, lib as L
ops.set_precision("bf16")
dt = torch.bfloat16
dev = "cuda"
N = 256
x = torch.randn(N, 256, 256, 8, device=dev).to(dt)
x[..., 3:] = 0
w = torch.randn(32, 3, 3, 3, device=dev) * 0.1
b = torch.zeros(32, device=dev)
g = ops.ConvGeom(3, 32, 3, 1, 1)
for rep in range(2):
    y = ops._conv_fwd_raw(x, w, b, g, L.ACT_NONE, dt)
    print(L.load().xmc_last_kernel().decode())
    torch.cuda.synchronize()
    y, yp = ops._conv_fwd_raw(x, w, b, g, L.ACT_NONE, dt, want_pool=True)
    torch.cuda.synchronize()
