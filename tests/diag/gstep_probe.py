"""What a joint 2B discriminator forward in the G step would save: time of netD(B, grad) + netD(B, no_grad) against netD(2B, grad),
and the backward of B against the backward of the second half of 2B (upper bound: here simply the backward of 2B / 2)."""
import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import importlib
ops = importlib.import_module("xmc-gan_amd.ops")
from parity_util import setup_cfg, build_product, X, DEV
ops.set_precision("bf16")
cfg, h = setup_cfg("df_gan_damsm_nomagp.yml", **{"IMG.SIZE": 256})
PG, PD = X.synth_params(X.gen_shapes(h), 3), X.synth_params(X.netd_shapes(h), 4)
netG, netD, _, _ = build_product(h, PG, PD)
for p in netD.parameters():
    p.requires_grad_(False)
B = 256
x2 = torch.randn(2 * B, 256, 256, 8, device=DEV).to(ops.act_dtype())
x2[..., 3:] = 0
xa, xb = x2[:B].contiguous(), x2[B:].contiguous()


def timed(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def sep():
    with torch.no_grad():
        netD(None, nhwc8=xa)
    xg = xb.clone().requires_grad_()
    return netD(None, nhwc8=xg), xg


def joint():
    xg = x2.clone().requires_grad_()
    return netD(None, nhwc8=xg), xg


print("forward  B(no grad) + B(grad):", round(timed(lambda: sep()), 3), "ms;  2B(grad):", round(timed(lambda: joint()), 3), "ms")
print("clone B:", round(timed(lambda: xb.clone()), 3), " clone 2B:", round(timed(lambda: x2.clone()), 3))


def fb(fn):
    f, xg = fn()
    f.float().sum().backward()


print("fwd+bwd  separate:", round(timed(lambda: fb(sep)), 3), "ms;  joint 2B (backward over all 2B):", round(timed(lambda: fb(joint)), 3), "ms")
