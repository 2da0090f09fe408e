"""Micro-driver for profiling one convolution shape in isolation (used with rocprofv3; not a test)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xmc_gan_amd import ops

def main():
    N, H, cin, cout, k, s, p, reps = [int(v) for v in sys.argv[1:9]]
    mode = sys.argv[9] if len(sys.argv) > 9 else "fwd"
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, H, H, ops.chan_pad(cin, torch.bfloat16), generator=g).to("cuda", torch.bfloat16).requires_grad_()
    w = torch.nn.Parameter((torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda())
    geom = ops.ConvGeom(cin, cout, k, s, p)
    y = ops.conv2d(x, w, None, geom)
    dy = torch.randn_like(y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if mode == "fwd":
            ops._conv_fwd_raw(x.detach(), w, None, geom, 0, torch.bfloat16)
        elif mode == "dgrad":
            ops._conv_dgrad_raw(dy, w, geom, (H, H), torch.bfloat16)
        else:
            ops._conv_wgrad_raw(x.detach(), dy, geom)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    fl = 2.0 * N * y.shape[1] * y.shape[2] * cin * cout * k * k
    print(f"{mode} N{N} {H}x{H} {cin}->{cout} k{k}s{s}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TF/s")

main()
