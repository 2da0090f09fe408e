"""Why does a kernel take longer inside the step than alone?  The same launch (discriminator block 0's conv_r[2] with its block end,
N = 256) over K rotating sets of buffers: K = 1 re-uses 1.8 GB, K = 24 walks 43 GB like the step does (cold TLB / MALL), and a
sustained run (3 s) shows what the power-capped clock does to it.
usage: python tests/diag/insitu_probe.py"""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xmc_gan_amd import ops, lib as L

dev = torch.device("cuda")
ops.set_precision("bf16")
dt = ops.act_dtype()
N, H, C = 256, 128, 64
w = torch.randn(C, C, 3, 3, device=dev) * 0.03
geom = ops.ConvGeom(C, C, 3, 1, 1)
al = torch.full((1,), 0.5, device=dev)
R = ops._conv_fwd_raw


def run(K, reps):
    sets = [(torch.randn(N, H, H, C, device=dev).to(dt), torch.randn(N, H, H, C, device=dev).to(dt)) for _ in range(K)]
    fn = lambda i: R(sets[i % K][0], w, None, geom, L.ACT_LRELU, dt, res=sets[i % K][1], alpha=al, round_act=True, want_pool=True, want_sign=True)
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    print(f"K = {K:3d} buffer sets, {reps:5d} launches back to back: {e0.elapsed_time(e1) / reps * 1000:8.1f} us / launch  {L.load().xmc_last_kernel().decode()}")
    del sets
    torch.cuda.empty_cache()


run(1, 20)
run(1, 4000)
run(24, 24)
run(24, 4000)
