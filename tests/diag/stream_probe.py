"""Times the HBM-bound pointwise kernels on full-size tensors through the C ABI and prints TB/s (diagnostic, not a test)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from xmc_gan_amd import lib as L, ops

bf = torch.bfloat16
st = lambda: ops._st()
p = ops._p


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (N, H, C) in ((512, 128, 64), (256, 256, 32), (256, 64, 128)):
    n = N * H * H * C
    dy = torch.randn(N, H, H, C, device="cuda").to(bf)
    ref = torch.randn(N, H, H, C, device="cuda").to(bf)
    g = torch.empty_like(dy)
    al = torch.full((1,), 0.3, device="cuda")
    dot = torch.zeros(1, device="cuda")
    ms = timeit(lambda: L.call("xmc_scale_mask_dot", p(dy), p(ref), p(al), p(g), p(dot), n, L.BF16, st()))
    print(f"scale_mask_dot N{N} {H}x{H}x{C}: {ms*1e3:7.1f} us  {3*n*2/ms/1e9:5.2f} TB/s")
    ps = [torch.randn(N, C, device="cuda") for _ in range(4)]
    y = torch.empty_like(dy)
    ms = timeit(lambda: L.call("xmc_affine2_act_fwd", p(dy), *[p(t) for t in ps], p(y), N, H * H, C, 0.2, L.BF16, st()))
    print(f"affine2_fwd    N{N} {H}x{H}x{C}: {ms*1e3:7.1f} us  {2*n*2/ms/1e9:5.2f} TB/s")
    red = torch.zeros(4, N, C, device="cuda")
    ms = timeit(lambda: L.call("xmc_affine2_act_bwd", p(dy), p(ref), *[p(t) for t in ps], p(y), *[p(red[i]) for i in range(4)], N, H * H, C, 0.2, L.BF16, st()))
    print(f"affine2_bwd    N{N} {H}x{H}x{C}: {ms*1e3:7.1f} us  {3*n*2/ms/1e9:5.2f} TB/s")
    ms = timeit(lambda: torch.add(dy, ref, out=g))
    print(f"torch add3     N{N} {H}x{H}x{C}: {ms*1e3:7.1f} us  {3*n*2/ms/1e9:5.2f} TB/s")

# GroupNorm / modulation / region attention of the concept blocks on their largest maps
for (N, H, C) in (((64, 128, 128),) if os.environ.get("XMC_PROBE_BIG") else ((64, 128, 128), (64, 64, 128), (64, 128, 64))):
    n = N * H * H * C
    x = torch.randn(N, H, H, C, device="cuda").to(bf)
    dy = torch.randn(N, H, H, C, device="cuda").to(bf)
    w = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    y, stats = ops._gn_fwd_raw(x, w, b, 16, 0.2, 1e-5)
    ms = timeit(lambda: ops._gn_fwd_raw(x, w, b, 16, 0.2, 1e-5))
    print(f"groupnorm fwd  N{N} {H}x{H}x{C}: {ms*1e3:7.1f} us  {3*n*2/ms/1e9:5.2f} TB/s (3 passes)")
    ms = timeit(lambda: ops._gn_bwd_raw(x, dy, w, b, stats, 16, 0.2))
    print(f"groupnorm bwd  N{N} {H}x{H}x{C}: {ms*1e3:7.1f} us  {5*n*2/ms/1e9:5.2f} TB/s (5 passes)")
