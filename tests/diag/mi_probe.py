"""Multi-image tiles of conv_wtile3.hip (8x8 maps, batch >= 256) against the same operator on a small batch slice (which the
dispatcher sends to the gather kernel): per-operator relative L2 differences."""
import sys, math, torch
sys.path.insert(0, "/root/repo")
import importlib
ops = importlib.import_module("xmc-gan_amd.ops")
L = importlib.import_module("xmc-gan_amd.lib")
DEV = "cuda"
ops.set_precision("bf16")
dt = ops.act_dtype()


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def kern():
    return L.load().xmc_last_kernel().decode()


g = torch.Generator().manual_seed(1)
N, S = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 8
for (cin, cout, k, s, p, H) in [(256, 256, 3, 1, 1, 8), (128, 512, 3, 1, 1, 8), (512, 512, 3, 1, 1, 8), (256, 512, 4, 2, 1, 16)]:
    x = torch.randn(N, H, H, cin, generator=g).to(DEV).to(dt)
    w = torch.nn.Parameter((torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).to(DEV))
    b = torch.nn.Parameter((torch.randn(cout, generator=g) * 0.1).to(DEV))
    geom = ops.ConvGeom(cin, cout, k, s, p)
    outs = []
    for sl in (slice(0, N), slice(0, S), slice(N - S, N)):
        xd = x[sl].clone().requires_grad_()
        y = ops.conv2d(xd, w, b, geom, act=L.ACT_LRELU)
        kf = kern()
        r = torch.randn(N, y.shape[1], y.shape[2], y.shape[3], generator=torch.Generator().manual_seed(7)).to(DEV).to(dt)[sl]
        w.grad = None
        (y.float() * r.float()).sum().backward()
        outs.append((y.detach(), xd.grad.detach(), kf))
    print(f"conv {cin}->{cout} k{k} s{s} H{H}: fwd kernel {outs[0][2]} vs {outs[1][2]}")
    print(f"   y  first {rel(outs[0][0][:S], outs[1][0]):.2e} last {rel(outs[0][0][N - S:], outs[2][0]):.2e}")
    print(f"   dx first {rel(outs[0][1][:S], outs[1][1]):.2e} last {rel(outs[0][1][N - S:], outs[2][1]):.2e}")

# fused upsample conv 8x8 -> 16x16
cin = cout = 256
x = torch.randn(N, 8, 8, cin, generator=g).to(DEV).to(dt)
w = torch.nn.Parameter((torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(DEV))
b = torch.nn.Parameter((torch.randn(cout, generator=g) * 0.1).to(DEV))
geom = ops.ConvGeom(cin, cout, 3, 1, 1)
outs = []
for sl in (slice(0, N), slice(0, S), slice(N - S, N)):
    xd = x[sl].clone().requires_grad_()
    y = ops.upconv3x3(xd, w, b, geom)
    kf = kern()
    r = torch.randn(N, 16, 16, cout, generator=torch.Generator().manual_seed(7)).to(DEV).to(dt)[sl]
    (y.float() * r.float()).sum().backward()
    outs.append((y.detach(), xd.grad.detach(), kf))
print(f"upconv: fwd kernel {outs[0][2]} vs {outs[1][2]}")
print(f"   y  first {rel(outs[0][0][:S], outs[1][0]):.2e} last {rel(outs[0][0][N - S:], outs[2][0]):.2e}")
print(f"   dx first {rel(outs[0][1][:S], outs[1][1]):.2e} last {rel(outs[0][1][N - S:], outs[2][1]):.2e}")

# discriminator block 16 -> 8 (256 -> 512) and 8x8 block end
from xmc_gan.model.df_gan import resD
for (ci, co, H) in [(256, 512, 16)]:
    torch.manual_seed(3)
    blk = resD(ci, co, downsample=True).to(DEV)
    with torch.no_grad():
        blk.gamma.fill_(0.37)
    x = torch.randn(N, H, H, ci, generator=g).to(DEV).to(dt)
    r = torch.randn(N, H // 2, H // 2, co, generator=g).to(DEV).to(dt)
    outs = []
    for sl in (slice(0, N), slice(0, S), slice(N - S, N)):
        xd = x[sl].clone().requires_grad_()
        blk.zero_grad()
        y = blk(xd)
        (y.float() * r[sl].float()).sum().backward()
        outs.append((y.detach(), xd.grad.detach(), blk.gamma.grad.clone()))
    print(f"resD {ci}->{co} H{H}:")
    print(f"   y  first {rel(outs[0][0][:S], outs[1][0]):.2e} last {rel(outs[0][0][N - S:], outs[2][0]):.2e}")
    print(f"   dx first {rel(outs[0][1][:S], outs[1][1]):.2e} last {rel(outs[0][1][N - S:], outs[2][1]):.2e}")
