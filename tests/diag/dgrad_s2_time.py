"""Timing aid (GPU): the data gradient of a discriminator block's 4x4 stride-2 convolution with the shortcut gradient as its
row-indexed residual (ResDBwdFn: dx = C0^T gh + 0.25 * up(dxp)), per block shape of the 256 px headline step.
usage: python tests/diag/dgrad_s2_time.py"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xmc_gan_amd import ops, lib as L

dev = torch.device("cuda")
ops.set_precision("bf16")
dt = ops.act_dtype()


def timeit(name, fn, gflop, gbytes, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"   {name:40s} {ms:8.3f} ms {gflop / ms:8.1f} TF/s {gbytes / ms:7.2f} TB/s  {L.load().xmc_last_kernel().decode()}")


N = 256
for (cin, cout, H) in [(32, 64, 256), (64, 128, 128), (128, 256, 64), (256, 512, 32), (512, 512, 16)]:
    OH = H // 2
    gh = torch.randn(N, OH, OH, cout, device=dev).to(dt)
    dxp = torch.randn(N, OH, OH, cin, device=dev).to(dt)
    w0 = torch.randn(cout, cin, 4, 4, device=dev) * 0.03
    g0 = ops.ConvGeom(cin, cout, 4, 2, 1)
    gf = 2.0 * N * OH * OH * cin * cout * 16 / 1e9
    gb = (gh.numel() + dxp.numel() + N * H * H * cin) * 2 / 1e9
    print(f"N{N} {cout}@{OH}x{OH} -> {cin}@{H}x{H}")
    timeit("dgrad", lambda: ops._conv_dgrad_raw(gh, w0, g0, (H, H), dt), gf, gb)
    timeit("dgrad + row residual", lambda: ops._conv_dgrad_raw(gh, w0, g0, (H, H), dt, res=dxp, res_rows=True, res_scale=0.25), gf, gb)
    del gh, dxp
    torch.cuda.empty_cache()
