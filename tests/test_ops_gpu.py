"""GPU parity tests of the individual HIP operators (through the C ABI) against plain PyTorch CPU math.

fp32 mode: f32 MFMA path, tolerance ~1e-4 (exact f32 products, different summation order).
bf16 mode: operands are pre-rounded to bf16 so the only differences are the f32-accumulate order and
the bf16 rounding of stored outputs (tolerance 2^-8 relative to the output scale).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from xmc_gan_amd import ops
    from xmc_gan_amd import lib as L
    import xmc_ref as X

DEV = "cuda"
MODES = ["fp32", "bf16"]


def rt(t, mode):
    """round-trip through the activation dtype (so CPU reference and kernel see identical operands)."""
    return t.to(torch.bfloat16).float() if mode == "bf16" else t.to(torch.float16).float() if mode == "f16" else t


def tol(mode, scale=1.0):
    if mode == "f16":        # the IEEE-half build of the same kernels: 2^-11 relative to the output scale
        return dict(rtol=3e-3, atol=3e-3 * scale)
    return dict(rtol=2e-2, atol=2e-2 * scale) if mode == "bf16" else dict(rtol=2e-4, atol=2e-4 * scale)


def to_nhwc(x, cpad, dtype):
    """CPU NCHW f32 -> device NHWC padded to cpad channels."""
    N, C, H, W = x.shape
    y = torch.zeros(N, H, W, cpad)
    y[..., :C] = x.permute(0, 2, 3, 1)
    return y.to(DEV, dtype).contiguous()


def from_nhwc(y, C):
    return y.float().cpu()[..., :C].permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # cin, cout, k, s, p, H, N
    (3, 32, 3, 1, 1, 16, 2),       # conv_img (Cin padded 3->8: four taps per K sub-step)
    (32, 64, 4, 2, 1, 16, 3),      # resD conv_r.0
    (64, 64, 3, 1, 1, 8, 2),       # resD conv_r.2 / G c2
    (32, 3, 3, 1, 1, 8, 2),        # conv_out (Cout padded 3->8)
    (96, 16, 3, 1, 1, 4, 5),       # joint_conv.0 style (odd tile tails: M=80)
    (16, 1, 4, 1, 0, 4, 6),        # joint_conv.2 (4x4 valid -> 1x1)
    (64, 128, 1, 1, 0, 8, 2),      # 1x1 shortcut, BN=128 tile
    (256, 256, 3, 1, 1, 4, 9),     # deep K, 128x128 tiles, M=144 (tail)
    (8, 8, 3, 1, 1, 32, 1),        # NCH=8 sized layer
    # halo-tile kernel (conv_tile.hip): unit-stride bf16 layers on >= 16x16 maps
    (64, 32, 3, 1, 1, 32, 2),      # 64-channel slab, BN=32, 8x32 tiles
    (32, 64, 3, 1, 1, 16, 3),      # 32-channel slab, BN=64, 16x16 tiles
    (128, 128, 3, 1, 1, 16, 2),    # two slabs, BN=128
    (64, 64, 4, 2, 1, 32, 2),      # forward on the gather kernel, dgrad = 4 parity classes of 2x2 taps on the tile kernel
    (32, 64, 1, 1, 0, 32, 2),      # 1x1 shortcut
    (192, 32, 3, 1, 1, 64, 1),     # three slabs, 64x64 map
    (128, 32, 3, 1, 1, 32, 2),     # the attention blocks' output convolution; its dgrad (32 -> 128): weights-resident kernel in two 64-channel halves
    # all-taps-per-tile weight-gradient kernel (conv_wgrad_tile.hip): Cin,Cout <= 64 on maps with W % 32 == 0
    (3, 32, 3, 1, 1, 32, 2),       # conv_img: Cin 3->8 (one padded 16-block)
    (32, 3, 3, 1, 1, 32, 2),       # conv_out: Cout 3->8
    (32, 32, 3, 1, 1, 32, 3),
    (64, 64, 3, 1, 1, 64, 1),      # row-reuse form (round 4): 3 (Cin block, kw) groups per wave, 2 Cout groups
    (32, 64, 3, 1, 1, 32, 2),      # ... 4 Cout groups x 2 group slices
    (64, 64, 3, 1, 1, 40, 2),      # H = 40: five tile rows (top / inner / bottom borders), W = 40 declines (not a multiple of 32)
    (64, 32, 1, 1, 0, 32, 2),
    # streaming kernel for 8-channel sources (conv_thin.hip): conv_img forward / conv_out dgrad on maps with H % 8 == 0, W % 32 == 0
    (3, 64, 3, 1, 1, 64, 1),       # 64 output channels, several tiles per image
    (64, 3, 3, 1, 1, 32, 3),       # its mirror: the dgrad has the 8-channel source
    # 256x128 tiles of the gather kernel: Cout % 128 == 0 and >= 65536 output pixels
    (128, 128, 3, 1, 1, 72, 13),   # M = 67392: last 256-row tile is partial
    (64, 128, 4, 2, 1, 128, 16),   # stride-2 forward; its dgrad = 4 parity classes (128 -> 64: a class's weights resident, ptile3 slab 128)
    (64, 128, 4, 2, 1, 64, 3),     # ... one tile column per class grid row: every tile touches the left and the right border
    (32, 256, 1, 1, 0, 64, 16),    # 1x1: a single K step
    # 256x256 tiles (8 waves) of the gather kernel: Cout % 256 == 0 and >= 65536 output pixels
    (128, 256, 4, 2, 1, 64, 72),   # stride-2 forward, M = 73728: partial last tile; dgrad: 4 classes with Cout 128
    (64, 512, 1, 1, 0, 32, 67),    # 1x1, two column tiles, M = 68608
    (512, 256, 4, 2, 1, 16, 8),    # small forward; its dgrad (512 -> ... no: 256-ch dy, 512-ch dx) stays on 128-wide tiles
    (256, 512, 4, 2, 1, 32, 258),  # dgrad: dy 512 ch -> dx 256 ch in 4 parity classes of 66048 pixels each on the 256x256 tile
    # weight gradient by kernel rows (conv_wgrad_row.hip): Cin, Cout % 128 == 0, W a power of two >= 16
    (128, 128, 3, 1, 1, 32, 2),    # two 32-pixel row segments per K step
    (128, 256, 3, 1, 1, 64, 1),    # one 64-pixel segment = one image row
    (256, 128, 3, 1, 1, 128, 1),   # image rows longer than a K step (segments start inside a row)
    (128, 128, 4, 2, 1, 32, 3),    # 4x4 stride 2: output rows of 16 pixels, 4 segments per step
    (128, 256, 4, 2, 1, 64, 2),    # output rows of 32
    (128, 128, 4, 2, 1, 128, 1),   # output rows of 64: 130 source pixels per segment
    (128, 128, 3, 1, 1, 8, 8),     # 8-pixel-wide map: 8 row segments per K step, pixel k+16 two segments on
    (256, 128, 3, 1, 1, 4, 16),    # 4-pixel-wide map: 16 segments per step
    (128, 256, 4, 2, 1, 16, 4),    # 4x4 stride 2 onto an 8x8 map (18 source pixels per segment)
    (128, 128, 4, 2, 1, 8, 16),    # ... onto a 4x4 map
    # all-taps weight gradient with 16x16 tiles (maps 16 pixels wide)
    (128, 128, 3, 1, 1, 16, 3),
    (256, 64, 3, 1, 1, 16, 2),     # 64-channel blocks over grid.y / grid.z
    # all-taps weight gradient of the 4x4 stride-2 layers with few channels (two column-parity planes in LDS)
    (32, 64, 4, 2, 1, 64, 2),      # resD block 0 shape: 64 co x 32 ci x 16 taps, 8 waves
    (16, 32, 4, 2, 1, 64, 3),      # 2 x 1 channel blocks, 4 waves
    # streaming 1x1 kernel (conv_thin.hip pw1x1): >= 16 k pixels, Cin <= 128, Cout <= 128
    (32, 64, 1, 1, 0, 64, 4),      # shortcut conv of resD block 0 (on the pooled input): one K step
    (64, 128, 1, 1, 0, 32, 16),    # two K steps, 8 row blocks; its dgrad: four K steps, 4 row blocks
    # deep-K steps of the gather kernel (KSUB 4) on the 4x4 maps: K >= 2048 with at most one workgroup per CU
    (256, 256, 3, 1, 1, 4, 128),   # 64x128 tiles, K = 2304 (9 taps x 256: the last step is partial)
    (512, 512, 3, 1, 1, 4, 512),   # 128x128 tiles, exactly 256 of them
    (512, 512, 4, 2, 1, 8, 256),   # 4x4 stride 2 onto 4x4, K = 8192; its dgrad: four 2x2-tap classes on 4x4 grids
    (768, 64, 3, 1, 1, 4, 95),     # the logit head's joint convolution at 3B-1 rows: 128x64 tiles, K = 6912
    # streaming 1x1 kernel with its weights in LDS (conv_thin.hip pw1x1w): Cin 128 / 256, >= 32 k pixels
    (128, 256, 1, 1, 0, 32, 40),   # shortcut of the 128 -> 256 block: one column slice; its dgrad (256 -> 128) the KS = 8 form
    (256, 512, 1, 1, 0, 16, 130),  # two column slices of 256; M = 33280 (last wave iteration partial); dgrad (Cin 512) on the gather kernel
    (256, 120, 1, 1, 0, 32, 33),   # Cout padded 120 -> 128
    # streamed-weights kernel on 8x8 maps (conv_wtile3.hip, tiles of four whole images); batch sizes at which it is chosen
    (256, 256, 3, 1, 1, 8, 512),   # 128-channel tiles (64x64 wave tiles), forward and data gradient
    (128, 512, 3, 1, 1, 8, 512),   # 256-channel tiles forward; the 128-channel data gradient stays on the gather kernel
    (256, 512, 4, 2, 1, 16, 512),  # 4x4 stride 2 onto 8x8 (space-to-depth patches of four images); dgrad = 4 classes of 2x2 taps on 8x8 grids
]


@pytest.mark.parametrize("mode", MODES + ["f16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, mode):
    cin, cout, k, s, p, H, N = case
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = rt(torch.randn(N, cin, H, H, generator=g), mode)
    w = rt(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k), mode)
    b = torch.randn(cout, generator=g) * 0.1
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    # Large cases run without the activation: among millions of pre-activations a few land within rounding of LeakyReLU's
    # kink, the GPU and CPU masks then differ in one element and that moves k*k*Cin entries of dx and Cin*k*k of dW by
    # 0.8*|r|*|x| -- the reference's kink, not a kernel error.  The fused activation is covered by the small cases.
    big = N * cout * H * H > ((1 << 14) if mode == "f16" else (1 << 20))      # f16: its tolerance is 8x tighter, the kink is not
    yr = F.conv2d(xr, wr, br, s, p) if big else F.leaky_relu(F.conv2d(xr, wr, br, s, p), 0.2)
    r = rt(torch.randn(yr.shape, generator=g), mode)
    (yr * r).sum().backward()

    geom = ops.ConvGeom(cin, cout, k, s, p)
    xd = to_nhwc(x, ops.chan_pad(cin, dt), dt).requires_grad_()
    wd = torch.nn.Parameter(w.to(DEV))
    bd = torch.nn.Parameter(b.to(DEV))
    y = ops.conv2d(xd, wd, bd, geom, act=L.ACT_NONE if big else L.ACT_LRELU)
    assert y.shape == (N, yr.shape[2], yr.shape[3], ops.pad_to(cout, 8))
    sc = yr.abs().max().item()
    torch.testing.assert_close(from_nhwc(y, cout), yr.detach(), **tol(mode, sc))
    if ops.pad_to(cout, 8) != cout:
        assert y[..., cout:].abs().max().item() == 0.0
    rd = to_nhwc(r, ops.pad_to(cout, 8), dt)
    (y.float() * rd.float()).sum().backward()
    torch.testing.assert_close(from_nhwc(xd.grad, cin), xr.grad, **tol(mode, xr.grad.abs().max().item()))
    torch.testing.assert_close(wd.grad.cpu(), wr.grad, **tol(mode, wr.grad.abs().max().item()))
    torch.testing.assert_close(bd.grad.cpu(), br.grad, **tol(mode, br.grad.abs().max().item()))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cin,cout,k,H,W,N", [(128, 128, 3, 4, 4, 3), (128, 128, 3, 32, 32, 2), (128, 128, 3, 20, 12, 2),
                                              (128, 64, 1, 16, 16, 3), (128, 64, 1, 7, 5, 1), (64, 64, 3, 8, 8, 2)])
def test_grouped_conv_fwd_dgrad_wgrad(cin, cout, k, H, W, N, mode):
    """nn.Conv2d(groups=16) of the attention-modulation blocks (df_concept_gan.py:146,267): the grouped kernel (bf16) or the
    dense kernels on the packed block-diagonal expansion (fp32 mode, shapes the grouped kernel declines) vs F.conv2d(groups=16)."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    G = 16
    g = torch.Generator().manual_seed(cin * 7 + cout + k + H * W)
    x = rt(torch.randn(N, cin, H, W, generator=g), mode)
    w = rt(torch.randn(cout, cin // G, k, k, generator=g) / math.sqrt(cin // G * k * k), mode)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = F.conv2d(xr, wr, None, 1, k // 2, groups=G)
    r = rt(torch.randn(yr.shape, generator=g), mode)
    (yr * r).sum().backward()
    geom = ops.ConvGeom(cin, cout, k, 1, k // 2, groups=G)
    xd = to_nhwc(x, cin, dt).requires_grad_()
    wd = torch.nn.Parameter(w.to(DEV))
    y = ops.conv2d(xd, wd, None, geom)
    if mode == "bf16":
        assert L.load().xmc_last_kernel().decode().startswith("gconv"), L.load().xmc_last_kernel()
    torch.testing.assert_close(from_nhwc(y, cout), yr.detach(), **tol(mode, yr.abs().max().item()))
    (y.float() * to_nhwc(r, cout, dt).float()).sum().backward()
    torch.testing.assert_close(from_nhwc(xd.grad, cin), xr.grad, **tol(mode, xr.grad.abs().max().item()))
    assert wd.grad.shape == wr.grad.shape
    torch.testing.assert_close(wd.grad.cpu(), wr.grad, **tol(mode, wr.grad.abs().max().item()))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cin,cout,H,N", [(64, 32, 16, 2), (32, 32, 32, 1), (128, 64, 8, 3), (256, 128, 4, 2), (8, 8, 16, 2),
                                           (128, 128, 16, 2), (256, 128, 8, 4),    # wide: weight gradient by kernel rows through the upsample
                                           (256, 256, 8, 512)])                    # 8x8 -> 16x16 at a batch that takes the multi-image tiles
def test_upsample_conv_fusion(cin, cout, H, N, mode):
    """conv3x3(interpolate(x, 2), w) + b computed on the low-resolution tensor with pre-summed 2x2 weights
    (df_gan.py:202 + 187): forward, dgrad (4x4-tap stride-2 gather), wgrad (through the fused upsample), bias grad;
    and up2(a) + gamma*b."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(cin * 7 + H)
    x = rt(torch.randn(N, cin, H, H, generator=g), mode)
    w = rt(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9), mode)
    b = torch.randn(cout, generator=g) * 0.1
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.conv2d(F.interpolate(xr, scale_factor=2), wr, br, 1, 1)
    r = rt(torch.randn(yr.shape, generator=g), mode)
    (yr * r).sum().backward()
    geom = ops.ConvGeom(cin, cout, 3, 1, 1)
    xd = to_nhwc(x, cin, dt).requires_grad_()
    wd, bd = torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV))
    y = ops.upconv3x3(xd, wd, bd, geom)
    # pre-summed weights are rounded once after the f32 sum: allow one extra bf16 ulp of the weight scale in bf16 mode
    t = tol(mode, yr.abs().max().item())
    if mode == "bf16":
        t = dict(rtol=3e-2, atol=3e-2 * yr.abs().max().item())
    torch.testing.assert_close(from_nhwc(y, cout), yr.detach(), **t)
    (y.float() * to_nhwc(r, cout, torch.float32)).sum().backward()
    tg = (lambda ref: dict(rtol=3e-2, atol=3e-2 * ref.abs().max().item())) if mode == "bf16" else (lambda ref: tol(mode, ref.abs().max().item()))
    torch.testing.assert_close(from_nhwc(xd.grad, cin), xr.grad, **tg(xr.grad))
    torch.testing.assert_close(wd.grad.cpu(), wr.grad, **tg(wr.grad))
    torch.testing.assert_close(bd.grad.cpu(), br.grad, **tg(br.grad))
    # up2(a) + gamma * b
    a = rt(torch.randn(N, cout, H, H, generator=g), mode)
    bb = rt(torch.randn(N, cout, 2 * H, 2 * H, generator=g), mode)
    gm = torch.tensor([0.6])
    ar, bbr, gmr = a.clone().requires_grad_(), bb.clone().requires_grad_(), gm.clone().requires_grad_()
    zr = F.interpolate(ar, scale_factor=2) + gmr * bbr
    (zr * r).sum().backward()
    ad, bbd = to_nhwc(a, cout, dt).requires_grad_(), to_nhwc(bb, cout, dt).requires_grad_()
    gmd = torch.nn.Parameter(gm.to(DEV))
    z = ops.axpby_up(ad, bbd, gmd)
    torch.testing.assert_close(from_nhwc(z, cout), zr.detach(), **tol(mode, 4.0))
    (z.float() * to_nhwc(r, cout, torch.float32)).sum().backward()
    torch.testing.assert_close(from_nhwc(ad.grad, cout), ar.grad, **tol(mode, 8.0))
    torch.testing.assert_close(from_nhwc(bbd.grad, cout), bbr.grad, **tol(mode, 4.0))
    torch.testing.assert_close(gmd.grad.cpu(), gmr.grad, **tol(mode, gmr.grad.abs().item() + 10.0))


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,H,W,N", [(64, 32, 128, 128, 2),      # generator block 6 (256 px): Cout 32, 4 x 32 low-res tiles, every border case
                                             (128, 64, 8, 64, 3),        # two Cin blocks over the grid, one row of tiles
                                             (256, 128, 12, 32, 2),      # Cin x Cout blocks over the grid, one column of tiles
                                             (256, 256, 16, 16, 5),      # 8 x 16 tiles of the 16-pixel-wide maps
                                             (64, 64, 24, 48, 1)])       # 16-wide tiles on a map that is not a power of two
def test_upsample_wgrad_low_resolution_kernel(cin, cout, H, W, N, mode):
    """the weight (and bias) gradient of conv3x3(interpolate(x, 2)) as 16 low-resolution products (conv_wgrad_up.hip) against
    autograd through F.interpolate + F.conv2d on the CPU (df_gan.py:202,217): same operands, f32 accumulation on both sides."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(cin + cout + H)
    x = rt(torch.randn(N, cin, H, W, generator=g), mode)
    dy = rt(torch.randn(N, cout, 2 * H, 2 * W, generator=g), mode)
    w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    b = torch.zeros(cout, requires_grad=True)
    (F.conv2d(F.interpolate(x, scale_factor=2), w, b, 1, 1) * dy).sum().backward()
    geom = ops.ConvGeom(cin, cout, 3, 1, 1)
    ops.new_iteration(DEV)
    gw, gb = ops._conv_wgrad_raw(to_nhwc(x, cin, dt), to_nhwc(dy, cout, dt), geom, up=True, want_bias=True)
    assert L.load().xmc_last_kernel().decode().startswith("wgrad_up_kernel"), L.load().xmc_last_kernel()
    for got, ref in ((gw, w.grad), (gb[:cout], b.grad)):
        err = (got.cpu() - ref).norm() / ref.norm()
        assert err < 5e-5, err                    # exact products of identical 16-bit operands; only the f32 summation order differs
    # border taps see the zero padding: compare tap by tap as well (a wrong class -> tap table would pass a norm over all taps
    # only by accident, but not this)
    for kh in range(3):
        for kw in range(3):
            e = (gw.cpu()[:, :, kh, kw] - w.grad[:, :, kh, kw]).abs().max() / w.grad[:, :, kh, kw].abs().max()
            assert e < 5e-4, (kh, kw, e)


@pytest.mark.parametrize("mode", MODES)
def test_linear_row_perm_and_mixed_dtype(mode):
    """proj_noise: f32 [B,100] -> NHWC [B,4,4,C] in the activation dtype via a row permutation."""
    ops.set_precision(mode)
    from xmc_gan.model.df_gan import nhwc_feature_perm
    g = torch.Generator().manual_seed(3)
    B, K, Cc = 5, 100, 16
    x = torch.randn(B, K, generator=g)
    w = torch.randn(Cc * 16, K, generator=g) / 10
    b = torch.randn(Cc * 16, generator=g)
    xr, wr, br = x.clone(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.linear(xr, wr, br).view(B, Cc, 4, 4)
    r = torch.randn(yr.shape, generator=g)
    (yr * r).sum().backward()
    geom = ops.ConvGeom(K, Cc * 16, 1, 1, 0, row_perm=nhwc_feature_perm(Cc))
    wd, bd = torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV))
    y = ops.linear(x.to(DEV), wd, bd, geom, out_dtype=ops.act_dtype()).view(B, 4, 4, Cc)
    torch.testing.assert_close(from_nhwc(y, Cc), yr.detach(), **tol(mode, yr.abs().max().item()))
    (y.float() * to_nhwc(r, Cc, torch.float32)).sum().backward()
    t = tol("fp32" if mode == "fp32" else "bf16", wr.grad.abs().max().item())
    torch.testing.assert_close(wd.grad.cpu(), wr.grad, **t)
    torch.testing.assert_close(bd.grad.cpu(), br.grad, **tol(mode, br.grad.abs().max().item()))


@pytest.mark.parametrize("mode", MODES)
def test_double_backward_gradient_penalty(mode):
    """grad-of-grad through conv(4x4,s2)+lrelu -> conv(3x3)+lrelu -> avgpool -> 1x1: the MA-GP pattern
    (train_gan.py:231-252) on a miniature discriminator."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(11)
    N, H = 3, 8
    x = rt(torch.randn(N, 8, H, H, generator=g), mode)
    w1 = rt(torch.randn(16, 8, 4, 4, generator=g) / 10, mode)
    w2 = rt(torch.randn(16, 16, 3, 3, generator=g) / 10, mode)
    w3 = rt(torch.randn(8, 16, 1, 1, generator=g) / 4, mode)
    gam = torch.tensor([0.7])

    def ref():
        xr = x.clone().requires_grad_()
        ws = [w.clone().requires_grad_() for w in (w1, w2, w3)]
        gm = gam.clone().requires_grad_()
        hdn = F.leaky_relu(F.conv2d(xr, ws[0], None, 2, 1), 0.2)
        h2 = F.leaky_relu(F.conv2d(hdn, ws[1], None, 1, 1), 0.2)
        out = F.conv2d(F.avg_pool2d(hdn + gm * h2, 2), ws[2])
        (gx,) = torch.autograd.grad(out.sum(), xr, create_graph=True)
        pen = (gx.reshape(N, -1).pow(2).sum(1).sqrt() ** 6).mean()
        grads = torch.autograd.grad(pen, ws + [gm])
        return pen.item(), gx.detach(), grads

    pen_r, gx_r, grads_r = ref()
    xd = to_nhwc(x, 8, dt).requires_grad_()
    ps = [torch.nn.Parameter(w.to(DEV)) for w in (w1, w2, w3)]
    gm = torch.nn.Parameter(gam.to(DEV))
    geoms = [ops.ConvGeom(8, 16, 4, 2, 1), ops.ConvGeom(16, 16, 3, 1, 1), ops.ConvGeom(16, 8, 1, 1, 0)]
    hdn = ops.conv2d(xd, ps[0], None, geoms[0], act=L.ACT_LRELU)
    h2 = ops.conv2d(hdn, ps[1], None, geoms[1], act=L.ACT_LRELU)
    out = ops.conv2d(ops.avgpool2(ops.axpby(hdn, h2, gm)), ps[2], None, geoms[2])
    with ops.no_wgrad():
        (gx,) = torch.autograd.grad(out, xd, torch.ones_like(out), create_graph=True)
    pen = (gx.float().reshape(N, -1).pow(2).sum(1).sqrt() ** 6).mean()
    grads = torch.autograd.grad(pen, ps + [gm])
    torch.testing.assert_close(from_nhwc(gx.detach(), 8), gx_r, **tol(mode, gx_r.abs().max().item()))
    assert abs(pen.item() - pen_r) <= (5e-2 if mode == "bf16" else 1e-3) * abs(pen_r)
    for a, b in zip(grads, grads_r):
        t = tol(mode, b.abs().max().item())
        if mode == "bf16":
            t = dict(rtol=6e-2, atol=6e-2 * b.abs().max().item())
        torch.testing.assert_close(a.cpu().reshape(b.shape), b, **t)


@pytest.mark.parametrize("mode", MODES)
def test_affine2_lrelu(mode):
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(5)
    N, Cc, H = 3, 24, 6
    x = rt(torch.randn(N, Cc, H, H, generator=g), mode)
    ps = [torch.randn(N, Cc, generator=g) for _ in range(4)]
    xr = x.clone().requires_grad_()
    pr = [p.clone().requires_grad_() for p in ps]
    e = lambda t: t[:, :, None, None]
    yr = F.leaky_relu(F.leaky_relu(xr * e(pr[0]) + e(pr[1]), 0.2) * e(pr[2]) + e(pr[3]), 0.2)
    r = rt(torch.randn(yr.shape, generator=g), mode)
    (yr * r).sum().backward()
    xd = to_nhwc(x, Cc, dt).requires_grad_()
    pd = [p.to(DEV).requires_grad_() for p in ps]
    y = ops.affine2_lrelu(xd, *pd)
    torch.testing.assert_close(from_nhwc(y, Cc), yr.detach(), **tol(mode, yr.abs().max().item()))
    (y.float() * to_nhwc(r, Cc, torch.float32)).sum().backward()
    torch.testing.assert_close(from_nhwc(xd.grad, Cc), xr.grad, **tol(mode, xr.grad.abs().max().item()))
    for a, b in zip(pd, pr):
        torch.testing.assert_close(a.grad.cpu(), b.grad, **tol(mode, b.grad.abs().max().item()))


@pytest.mark.parametrize("mode", MODES)
def test_pool_resample_convert(mode):
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(6)
    x = rt(torch.randn(2, 16, 8, 8, generator=g), mode)
    xd = to_nhwc(x, 16, dt).requires_grad_()
    xr = x.clone().requires_grad_()
    # avgpool2 / upsample2 / global pool, forward and backward
    for fn_d, fn_r in ((ops.avgpool2, lambda t: F.avg_pool2d(t, 2)),
                       (ops.upsample2, lambda t: F.interpolate(t, scale_factor=2)),
                       (lambda t: ops.lrelu(t), lambda t: F.leaky_relu(t, 0.2))):
        xd.grad = None; xr.grad = None
        yd, yr = fn_d(xd), fn_r(xr)
        torch.testing.assert_close(from_nhwc(yd, 16), yr.detach(), **tol(mode, 3.0))
        r = rt(torch.randn(yr.shape, generator=g), mode)
        (yd.float() * to_nhwc(r, 16, torch.float32)).sum().backward()
        (yr * r).sum().backward()
        torch.testing.assert_close(from_nhwc(xd.grad, 16), xr.grad, **tol(mode, 3.0))
    xd.grad = None; xr.grad = None
    x4 = xd[:, :4, :4, :].contiguous()
    yd = ops.global_avgpool(x4)
    yr = F.avg_pool2d(xr[:, :, :4, :4], 4).view(2, 16)
    torch.testing.assert_close(yd.cpu(), yr.detach(), **tol(mode))
    r = torch.randn(2, 16, generator=g)
    (yd * r.to(DEV)).sum().backward(); (yr * r).sum().backward()
    torch.testing.assert_close(from_nhwc(xd.grad, 16), xr.grad, **tol(mode))
    # image boundary converters
    img = torch.rand(3, 3, 8, 8, generator=g) * 2 - 1
    imd = img.to(DEV).requires_grad_()
    z = ops.to_nhwc8(imd)
    assert z.shape == (3, 8, 8, 8) and z[..., 3:].abs().max().item() == 0
    back = ops.to_nchw(z, 3)
    torch.testing.assert_close(back.cpu(), rt(img, mode), rtol=0, atol=0)
    back.backward(torch.ones_like(back))
    torch.testing.assert_close(imd.grad.cpu(), torch.ones_like(img), rtol=0, atol=0)


@pytest.mark.parametrize("mode", MODES)
def test_axpby_scale_dot_hinge(mode):
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(8)
    a = rt(torch.randn(2, 4, 4, 16, generator=g), mode)
    b = rt(torch.randn(2, 4, 4, 16, generator=g), mode)
    al = torch.tensor([0.37])
    ar, br_, alr = a.clone().requires_grad_(), b.clone().requires_grad_(), al.clone().requires_grad_()
    yr = ar + alr * br_
    r = rt(torch.randn(yr.shape, generator=g), mode)
    (yr * r).sum().backward()
    ad, bd = a.to(DEV, dt).requires_grad_(), b.to(DEV, dt).requires_grad_()
    ald = torch.nn.Parameter(al.to(DEV))
    y = ops.axpby(ad, bd, ald)
    torch.testing.assert_close(y.float().cpu(), yr.detach(), **tol(mode, 3.0))
    (y.float() * r.to(DEV)).sum().backward()
    torch.testing.assert_close(ad.grad.float().cpu(), ar.grad, **tol(mode, 3.0))
    torch.testing.assert_close(bd.grad.float().cpu(), br_.grad, **tol(mode, 3.0))
    torch.testing.assert_close(ald.grad.cpu(), alr.grad, **tol(mode, 30.0))
    # hinge on channel 0 of a padded logit tensor, read through a strided view
    lg = rt(torch.randn(7, 1, 1, 8, generator=g), mode)
    lgd = lg.to(DEV, dt).requires_grad_()
    view = lgd[..., :1].permute(0, 3, 1, 2)
    for sign in (-1.0, 1.0):
        lgd.grad = None
        lr_ = lg[..., 0].clone().requires_grad_()
        ref = F.relu(1.0 + sign * lr_).mean()
        ref.backward()
        out = ops.hinge(view, sign)
        assert abs(out.item() - ref.item()) < 1e-5
        out.backward()
        torch.testing.assert_close(lgd.grad.float().cpu()[..., 0], lr_.grad, **tol(mode))
        assert lgd.grad[..., 1:].abs().max().item() == 0


@pytest.mark.parametrize("n,D", [(8, 256), (5, 48), (64, 512), (200, 256),
                                 (256, 256), (2048, 256), (2048, 512)])     # one GPU's batch; 8 x 256 all-gathered rows (BASELINE config 5)
@pytest.mark.parametrize("labels_kind", ["identity", "global_adaptive", "global_smooth"])
def test_contrastive_head(n, D, labels_kind):
    """fused cosine-similarity + symmetric InfoNCE vs the oracle (train_gan.py:85-139); f32, logits/loss 1e-3 rel."""
    g = torch.Generator().manual_seed(n * 7 + D)
    a = torch.randn(n, D, generator=g)
    b = a * 0.7 + torch.randn(n, D, generator=g)
    sent = torch.randn(max(n // 4, 1), D, generator=g).repeat(4, 1)[:n]
    sent = torch.cat([sent, torch.randn(n - sent.size(0), D, generator=g)]) + 0.3 * torch.randn(n, D, generator=g)
    if labels_kind == "identity":
        labels, bg, sg = X.make_labels(n, sent, False), False, 0.0
    elif labels_kind == "global_adaptive":
        labels, bg, sg = X.make_labels(n, sent, True, 0.0), True, 0.0
    else:
        labels, bg, sg = X.make_labels(n, sent, True, 0.5), True, 0.5
    ar, br_ = a.clone().requires_grad_(), b.clone().requires_grad_()
    ref = X.contrastive_loss(ar, br_, labels, bg, sg)
    ref.backward()
    if not bg:
        inv_np = None
    elif sg == 0.0:
        inv_np = torch.full((n,), 0.5)
    else:
        inv_np = 1.0 / (labels > 0).sum(1).float()
    ad, bd = a.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    lab_d = None if labels_kind == "identity" else labels.to(DEV).contiguous()
    out = ops.contrastive(ad, bd, lab_d, None if inv_np is None else inv_np.to(DEV))
    assert abs(out.item() - ref.item()) <= 1e-4 * abs(ref.item()) + 1e-5
    (out * 1.7).backward()
    torch.testing.assert_close(ad.grad.cpu(), 1.7 * ar.grad, rtol=2e-3, atol=2e-6)
    torch.testing.assert_close(bd.grad.cpu(), 1.7 * br_.grad, rtol=2e-3, atol=2e-6)
    # explicit identity labels must give the same result as the labels=None fast path
    if labels_kind == "identity":
        out2 = ops.contrastive(ad.detach(), bd.detach(), labels.to(DEV).contiguous(), None)
        assert abs(out2.item() - out.item()) < 1e-5 * abs(out.item())


def test_rejects_cpu_tensors_and_bad_shapes():
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 4, 4, 8), torch.nn.Parameter(torch.zeros(8, 8, 3, 3)), None, ops.ConvGeom(8, 8, 3, 1, 1))
    d = L.ConvDesc()
    import ctypes as C
    assert L.load().xmc_conv_igemm(C.byref(d), None) < 0          # null pointers -> XMC_EINVAL, nothing launched


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cin,cout,H", [(32, 64, 32), (64, 64, 16), (8, 16, 64)])
def test_fused_discriminator_block_equals_composed_block(cin, cout, H, mode):
    """ops.ResDFn (one first-order node per resD block: masks, d(gamma) and the pooled shortcut gradient folded into the
    convolution epilogues) against the same block built from the fine-grained Functions (df_gan.py:269-291)."""
    from xmc_gan.model.df_gan import resD
    ops.set_precision(mode)
    dt = ops.act_dtype()
    torch.manual_seed(cin + H)
    blk = resD(cin, cout, downsample=True).to(DEV)
    with torch.no_grad():
        blk.gamma.fill_(0.37)
    x0 = torch.randn(3, H, H, ops.chan_pad(cin, dt), device=DEV).to(dt)
    r = torch.randn(3, H // 2, H // 2, ops.pad_to(cout, 8), device=DEV).to(dt)
    got = {}
    # "fused": the block keeps the SIGN BITS of its residual branch (XmcConvDesc.sign_bits) and gets d(gamma) from the dot in the
    # data-gradient epilogue; "fused_values": the branch tensor itself (what ops.second_order() selects for MA-GP)
    for name in ("fused", "fused_values", "composed"):
        blk.zero_grad()
        x = x0.clone().requires_grad_()
        with (ops.composable() if name == "composed" else ops.second_order(name == "fused_values")):
            y = blk(x)
        (y.float() * r.float()).sum().backward()
        got[name] = [y.detach().float(), x.grad.float()] + [p.grad.clone() if p.grad is not None else None for p in blk.parameters()]
    names = ["y", "dx"] + [n for n, _ in blk.named_parameters()]
    for n, a, b, c in zip(names, got["fused"], got["composed"], got["fused_values"]):
        assert (a is None) == (b is None) == (c is None), n
        if a is None:
            continue
        sc = b.abs().max().item() + 1e-12
        torch.testing.assert_close(c, b, rtol=2e-2 if mode == "bf16" else 1e-4, atol=(2e-2 if mode == "bf16" else 1e-4) * sc, msg=lambda m: f"{n} (values): {m}")
        # bf16: the composed path rounds dx of each branch to bf16 before adding them, the fused path adds in f32
        torch.testing.assert_close(a, b, rtol=2e-2 if mode == "bf16" else 1e-4, atol=(2e-2 if mode == "bf16" else 1e-4) * sc, msg=lambda m: f"{n}: {m}")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cin,cout,H", [(32, 64, 32), (64, 64, 16), (8, 16, 64)])
def test_fused_discriminator_block_double_backward_equals_composed_block(cin, cout, H, mode):
    """The MA-GP pattern (train_gan.py:231-252) on one block: g = d<y, r>/dx with create_graph, then the gradient of a
    non-linear function of g with respect to the block's parameters AND r.  Fused path: ops.ResDFn -> ops.ResDBwdFn (whose
    backward is the linearised forward of the block); reference: the block composed from the fine-grained Functions."""
    from xmc_gan.model.df_gan import resD
    ops.set_precision(mode)
    dt = ops.act_dtype()
    torch.manual_seed(cin + H + 1)
    blk = resD(cin, cout, downsample=True).to(DEV)
    with torch.no_grad():
        blk.gamma.fill_(0.41)
    x0 = torch.randn(3, H, H, ops.chan_pad(cin, dt), device=DEV).to(dt)
    r0 = torch.randn(3, H // 2, H // 2, ops.pad_to(cout, 8), device=DEV).to(dt)
    t = torch.randn(3, H, H, ops.chan_pad(cin, dt), device=DEV)
    got = {}
    for name in ("fused", "composed"):
        blk.zero_grad()
        x = x0.clone().requires_grad_()
        r = r0.clone().requires_grad_()
        with (ops.composable() if name == "composed" else ops.second_order()):
            y = blk(x)
        with ops.no_wgrad():
            g, = torch.autograd.grad(y, x, grad_outputs=r, create_graph=True)
        loss = (g.float() * t).sum() + 0.5 * (g.float() ** 2).sum()
        loss.backward()
        got[name] = [g.detach().float(), r.grad.float()] + [p.grad.clone() if p.grad is not None else None for p in blk.parameters()]
    names = ["g", "d/dr"] + [n for n, _ in blk.named_parameters()]
    for n, a, b in zip(names, got["fused"], got["composed"]):
        assert (a is None) == (b is None), n
        if a is None:
            continue
        sc = b.abs().max().item() + 1e-12
        torch.testing.assert_close(a, b, rtol=3e-2 if mode == "bf16" else 2e-4, atol=(3e-2 if mode == "bf16" else 2e-4) * sc, msg=lambda m: f"{n}: {m}")


@pytest.mark.parametrize("N,H,C", [(2, 16, 64), (8, 32, 256), (4, 64, 128), (16, 16, 512), (2, 128, 64), (3, 24, 40), (64, 32, 256), (32, 32, 512),
                                   (512, 8, 256), (512, 8, 512)])          # 8x8 maps: multi-image tiles, 128- and 256-channel
def test_conv_sign_bits_and_gradient_dot_against_the_stored_branch(N, H, C):
    """XmcConvDesc.sign_bits / .dot (include/xmc_gan_hip.h), the two epilogue options a discriminator block's first-order backward
    runs on: the bits must be exactly `branch > 0` of the branch the kernel would otherwise store, the block output must not change
    by a bit, and the dot must equal <C2^T g, h1> of the unmasked data gradient.  Shapes pick different kernels (thin / weights-
    resident / streamed-weights tiles); the names are printed with -s."""
    ops.set_precision("bf16")
    dt = ops.act_dtype()
    g = torch.Generator(device="cpu").manual_seed(N * 1000 + H + C)
    Cp = ops.pad_to(C, 8)
    h1 = torch.nn.functional.leaky_relu(torch.randn(N, H, H, Cp, generator=g), 0.2).to(DEV).to(dt)
    sc = torch.randn(N, H, H, Cp, generator=g).to(DEV).to(dt)
    w2 = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(DEV)
    if Cp != C:
        h1[..., C:] = 0
        sc[..., C:] = 0
    al = torch.tensor([0.63], device=DEV)
    g2 = ops.ConvGeom(C, C, 3, 1, 1)
    pool = H % 2 == 0
    out_v, branch, *pv = ops._conv_fwd_raw(h1, w2, None, g2, L.ACT_LRELU, dt, res=sc, alpha=al, want2=True, want_pool=pool, round_act=True)
    k_v = L.load().xmc_last_kernel().decode()
    out_s, bits, *ps = ops._conv_fwd_raw(h1, w2, None, g2, L.ACT_LRELU, dt, res=sc, alpha=al, want_sign=True, want_pool=pool, round_act=True)
    k_s = L.load().xmc_last_kernel().decode()
    print(f"forward: {k_v} / {k_s}")
    assert torch.equal(out_v, out_s) and (not pool or torch.equal(pv[0], ps[0]))
    assert bits.dtype == torch.uint8 and tuple(bits.shape) == (N, H, H, Cp // 8)
    want = ((branch.float() > 0).view(N, H, H, Cp // 8, 8).to(torch.int32) << torch.arange(8, device=DEV, dtype=torch.int32)).sum(-1)
    assert torch.equal(bits.to(torch.int32), want)
    # the mask pass from the bits == the mask pass from the branch values
    dout = torch.randn(N, H, H, Cp, generator=g).to(DEV).to(dt)
    gr_s = torch.empty_like(dout)
    L.call("xmc_signmask_apply", dout.data_ptr(), bits.data_ptr(), gr_s.data_ptr(), dout.numel(), 0.2, ops._code(dt), ops._st())
    gr_v = torch.where(branch.float() > 0, dout.float(), 0.2 * dout.float()).to(dt)
    assert torch.equal(gr_s, gr_v)
    # data gradient with gamma and the dot in its epilogue
    u = ops._conv_dgrad_raw(gr_s, w2, g2, (H, H), dt)                                  # C2^T g, unmasked, rounded to bf16
    ref = ops._conv_dgrad_raw(gr_s, w2, g2, (H, H), dt, mask=h1, alpha=al)
    dgam = torch.zeros(1, device=DEV)
    got = ops._conv_dgrad_raw(gr_s, w2, g2, (H, H), dt, mask=h1, alpha=al, dot=dgam)
    print(f"data gradient: {L.load().xmc_last_kernel().decode()}")
    assert torch.equal(got, ref)
    want_dot = (u.double() * h1.double()).sum().item()
    scale = (u.double() * h1.double()).abs().sum().item()
    assert abs(dgam.item() - want_dot) <= 2e-3 * scale / (u.numel() ** 0.5) * 30 + 1e-6 * scale, (dgam.item(), want_dot, scale)


@pytest.mark.parametrize("N", [512, 256])
def test_multi_image_tiles_equal_the_small_batch_kernels(N):
    """conv_wtile3.hip's tiles of four whole 8x8 images (chosen from batch 256 up) against the SAME operator on 8-sample slices,
    which the dispatcher sends to the gather kernel: same bf16 operands, f32 accumulation in another order -> relative L2 <= 5e-4
    (measured 1e-5 .. 1e-4; a wrong halo or image offset would be O(1)).  Covers 3x3, the 4x4 stride-2 forward (space-to-depth
    patches) and its four-class data gradient, the fused upsample conv and a whole discriminator block with its fused epilogues."""
    from xmc_gan.model.df_gan import resD
    ops.set_precision("bf16")
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(N)
    S = 8
    slices = (slice(0, N), slice(0, S), slice(N - S, N))
    rel = lambda a, b: ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()
    last = lambda: L.load().xmc_last_kernel().decode()

    def check(name, run, x, r, want_kernel=True, tol=5e-4):
        outs = []
        for sl in slices:
            xd = x[sl].clone().requires_grad_()
            y, kf = run(xd)
            (y.float() * r[sl].float()).sum().backward()
            outs.append((y.detach(), xd.grad.detach(), kf))
        if want_kernel:
            assert outs[0][2].startswith("wtile3_kernel") and not outs[1][2].startswith("wtile3_kernel"), (name, outs[0][2], outs[1][2])
        for which, a, b in (("y", 0, 0), ("dx", 1, 1)):
            e0, e1 = rel(outs[0][a][:S], outs[1][b]), rel(outs[0][a][N - S:], outs[2][b])
            assert e0 < tol and e1 < tol, (name, which, e0, e1)

    for (cin, cout, k, s, p, H) in [(512, 512, 3, 1, 1, 8), (128, 512, 3, 1, 1, 8), (256, 512, 4, 2, 1, 16)]:
        x = torch.randn(N, H, H, cin, generator=g).to(DEV).to(dt)
        w = torch.nn.Parameter((torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).to(DEV))
        b = torch.nn.Parameter((torch.randn(cout, generator=g) * 0.1).to(DEV))
        geom = ops.ConvGeom(cin, cout, k, s, p)
        Ho = (H + 2 * p - k) // s + 1
        r = torch.randn(N, Ho, Ho, cout, generator=g).to(DEV).to(dt)

        def run(xd, w=w, b=b, geom=geom):
            y = ops.conv2d(xd, w, b, geom, act=L.ACT_LRELU)
            return y, last()
        check(f"conv {cin}->{cout} k{k} s{s}", run, x, r)
    x = torch.randn(N, 8, 8, 256, generator=g).to(DEV).to(dt)
    w = torch.nn.Parameter((torch.randn(256, 256, 3, 3, generator=g) / math.sqrt(256 * 9)).to(DEV))
    b = torch.nn.Parameter((torch.randn(256, generator=g) * 0.1).to(DEV))
    r = torch.randn(N, 16, 16, 256, generator=g).to(DEV).to(dt)

    def run_up(xd):
        y = ops.upconv3x3(xd, w, b, ops.ConvGeom(256, 256, 3, 1, 1))
        return y, last()
    check("upconv 256->256 8x8 -> 16x16", run_up, x, r)
    torch.manual_seed(3)
    blk = resD(256, 512, downsample=True).to(DEV)
    with torch.no_grad():
        blk.gamma.fill_(0.37)
    x = torch.randn(N, 16, 16, 256, generator=g).to(DEV).to(dt)
    r = torch.randn(N, 8, 8, 512, generator=g).to(DEV).to(dt)
    # a whole block: its inner LeakyReLU mask flips where a pre-activation sits within that 1e-4 of zero (0.8 |g| per element)
    check("resD 256->512 16 -> 8", lambda xd: (blk(xd), ""), x, r, want_kernel=False, tol=5e-3)


@pytest.mark.parametrize("N,H,cin,cout", [(16, 32, 32, 64), (64, 32, 64, 128), (40, 32, 128, 256), (130, 16, 256, 512), (2, 8, 32, 64)])
def test_shortcut_data_gradient_writes_the_masked_gradient_as_a_by_product(N, H, cin, cout):
    """xmc_conv_pw1x1_masked_src (include/xmc_gan_hip.h): the streaming 1x1 kernels, reading `dout` as the source of the learned
    shortcut's data gradient, also write dout x LeakyReLU'(sign bits) -- both outputs bit-equal to the two separate launches
    (xmc_conv_igemm, xmc_signmask_apply).  Shapes: registers-resident weights (Cin of the data gradient 64 / 128), weights in LDS
    (256 / 512: the 512-channel source is declined by the streaming kernels and takes the fallback), and a map too small for them."""
    ops.set_precision("bf16")
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(N + cin)
    dout = torch.randn(N, H, H, cout, generator=g).to(DEV).to(dt)
    bits = torch.randint(0, 256, (N, H, H, cout // 8), generator=g, dtype=torch.int64).to(torch.uint8).to(DEV)
    ws = (torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)).to(DEV)
    gs = ops.ConvGeom(cin, cout, 1, 1, 0)
    dx_ref = ops._conv_dgrad_raw(dout, ws, gs, (H, H), dt)
    gr_ref = torch.empty_like(dout)
    L.call("xmc_signmask_apply", dout.data_ptr(), bits.data_ptr(), gr_ref.data_ptr(), dout.numel(), 0.2, ops._code(dt), ops._st())
    dx, gr = ops._conv_dgrad_raw(dout, ws, gs, (H, H), dt, src_bits=bits)
    print(L.load().xmc_last_kernel().decode())
    assert torch.equal(gr, gr_ref)
    assert torch.equal(dx, dx_ref)


@pytest.mark.parametrize("mode", ["f16", "bf16"])
@pytest.mark.parametrize("rows", [8, 767])
def test_pair_convolution_reaches_f32_grade_on_the_16_bit_pipeline(rows, mode):
    """ops.PairConvFn (COND_DNET's joint_conv.0 on the precise trunk, df_gan.py:157,170-174): an f32 [rows,4,4,768] map through the
    3x3 convolution to 64 channels + LeakyReLU with both operands as 16-bit hi + lo pairs, three launches accumulating in f32.
    Against an f64 evaluation: 2^-20-grade (the dropped lo x lo product and the f32 accumulation), where one 16-bit launch is
    2^-12 (f16) / 2^-9 (bf16); gradients are the 16-bit layer's."""
    ops.set_precision(mode)
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 4, 4, 768, generator=g)
    w = torch.randn(64, 768, 3, 3, generator=g) * math.sqrt(2.0 / (768 * 9))
    gs = ops.ConvGeom(768, 64, 3, 1, 1)
    wd = torch.nn.Parameter(w.to(DEV))
    xd = x.to(DEV).requires_grad_()
    y = ops.pair_conv2d(xd, wd, gs, L.ACT_LRELU)
    ref = F.leaky_relu(F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, 1, 1), 0.2).permute(0, 2, 3, 1)
    e = rel_l2(y.detach().float().cpu(), ref.float())
    y1 = ops.conv2d(xd.detach().to(ops.act_dtype()), wd, None, gs, L.ACT_LRELU, torch.float32)
    e1 = rel_l2(y1.float().cpu(), ref.float())
    print(f"\n[{mode} rows {rows}] pair convolution vs f64: {e:.2e}; one 16-bit launch: {e1:.2e}")
    assert e <= (3e-6 if mode == "f16" else 4e-5) and e1 >= 20 * e
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.to(DEV))
    xr = x.double().permute(0, 3, 1, 2).requires_grad_()
    wr = w.double().requires_grad_()
    F.leaky_relu(F.conv2d(xr, wr, None, 1, 1), 0.2).backward(dy.double().permute(0, 3, 1, 2))
    ulp = 2.0 ** -11 if mode == "f16" else 2.0 ** -8
    assert rel_l2(xd.grad.float().cpu(), xr.grad.permute(0, 2, 3, 1).float()) <= 2 * ulp
    assert rel_l2(wd.grad.float().cpu(), wr.grad.float()) <= 2 * ulp
    ops.set_precision("bf16")


@pytest.mark.parametrize("mode", ["f16", "bf16"])
@pytest.mark.parametrize("N,H,cin,cout,kernel", [(16, 32, 64, 128, "pw1x1_kernel<2, 8, false, true>"), (32, 32, 128, 256, "pw1x1w_kernel<4, false, true>"),
                                                 (128, 16, 256, 512, "pw1x1w_kernel<8, false, true>"), (8, 16, 64, 128, "igemm"), (4, 32, 8, 16, "igemm")])
def test_learned_shortcut_on_a_weight_pair(N, H, cin, cout, kernel, mode):
    """xmc_conv_pw1x1_split (ABI 12, XmcConvDesc.wpk_lo): conv_s with its f32 weights as the 16-bit pair round16(w) + round16(w - round16(w)).
    A weight's rounding error is the SAME for every pixel, so it survives a mean over pixels, where the rounding of the stored
    16-bit outputs averages out: per output channel, |mean over pixels of (y - y_exact)| / rms(y_exact) is ~2^-9 (bf16) / 2^-12
    (f16) x |mean x| / rms x with plain weights and at the noise floor of the output rounding with the pair.  Shapes: the three
    streaming instantiations, and two the streaming kernels decline (few pixels / narrow layers: exact-f32 MFMA on the widened
    input).  The half build must not flush the pair's low half (subnormal for |w| < 0.12): the f16 bar would fail by 30x."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(N + cin)
    x = (torch.randn(N, H, H, cin, generator=g).abs() + 0.25).to(dt)           # a non-zero mean per channel
    w = (torch.randn(cout, cin, 1, 1, generator=g) * math.sqrt(2.0 / cin))
    b = torch.randn(cout, generator=g) * 0.1
    gs = ops.ConvGeom(cin, cout, 1, 1, 0)
    wd, bd = torch.nn.Parameter(w.to(DEV)), b.to(DEV)
    y = ops._conv1x1_pair_raw(x.to(DEV), wd, bd, gs, dt)
    kn = L.load().xmc_last_kernel().decode()
    y_plain = ops._conv_fwd_raw(x.to(DEV), wd, bd, gs, L.ACT_NONE, dt)
    ref = (x.double().reshape(-1, cin) @ w.double().reshape(cout, cin).t() + b.double())
    rms = ref.pow(2).mean(0).sqrt()
    bias_of = lambda t: ((t.double().cpu().reshape(-1, cout) - ref).mean(0).abs() / rms).max().item()
    e_pair, e_plain = bias_of(y), bias_of(y_plain)
    ulp = 2.0 ** -12 if mode == "f16" else 2.0 ** -9
    floor = ulp / math.sqrt(N * H * H) * 4
    print(f"\n[{mode} {cin}->{cout} N{N} {H}x{H}] {kn}: coherent error pair {e_pair:.2e}, plain weights {e_plain:.2e} (noise floor ~{floor:.1e})")
    assert kernel in kn, kn
    assert e_pair <= max(3 * floor, ulp / 30), (e_pair, e_plain)
    assert rel_l2(y.float().cpu(), ref.float().reshape(y.shape)) <= 1.2 * ulp        # element-wise: the rounding of the stored output
    ops.set_precision("bf16")


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("N,H,W,cin,cout,fused", [(3, 32, 64, 64, 64, True), (2, 128, 128, 64, 64, True), (5, 16, 32, 32, 64, True),
                                                  (2, 16, 16, 128, 128, False), (2, 24, 24, 64, 64, False)])
def test_sign_mask_applied_while_the_gradient_is_staged(N, H, W, cin, cout, fused, mode):
    """XmcConvDesc.mask_bits (xmc_conv_ptile_bits / xmc_conv_wgrad_bits, include/xmc_gan_hip.h): the consumers of a block's masked
    gradient read dout and its sign bytes and never see the masked tensor in memory.  The data gradient (with the epilogue options
    the block uses) must equal the two-launch form bit for bit; the weight gradient sums the same products in the same kernel, so
    only the order of its final atomics differs (rel L2 <= 1e-6).  Shapes the staging kernels decline (`fused` False) take the mask
    pass once, shared by both consumers."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(N * 7 + H + cin)
    dout = torch.randn(N, H, W, cout, generator=g).to(DEV).to(dt)
    bits = torch.randint(0, 256, (N, H, W, cout // 8), generator=g, dtype=torch.int64).to(torch.uint8).to(DEV)
    h1 = torch.randn(N, H, W, cin, generator=g).to(DEV).to(dt)
    w2 = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).to(DEV)
    g2 = ops.ConvGeom(cin, cout, 3, 1, 1)
    al = torch.tensor([0.41], device=DEV)
    gr = torch.empty_like(dout)
    L.call("xmc_signmask_apply", dout.data_ptr(), bits.data_ptr(), gr.data_ptr(), dout.numel(), 0.2, ops._code(dt), ops._st())
    dot_ref, dot_got = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    dx_ref = ops._conv_dgrad_raw(gr, w2, g2, (H, W), dt, mask=h1, alpha=al, dot=dot_ref)
    dw_ref = ops._conv_wgrad_raw(h1, gr, g2, scale=al)
    st = ops._StagedMask(dout, bits)
    dx = ops._conv_dgrad_raw(st, w2, g2, (H, W), dt, mask=h1, alpha=al, dot=dot_got)
    kd = L.load().xmc_last_kernel().decode()
    dw = ops._conv_wgrad_raw(h1, st, g2, scale=al)
    kw = L.load().xmc_last_kernel().decode()
    print(kd, "/", kw)
    assert (st._full is None) == fused, (kd, kw)
    assert torch.equal(dx, dx_ref)
    assert abs(dot_got.item() - dot_ref.item()) <= 1e-5 * abs(dot_ref.item()) + 1e-4
    assert rel_l2(dw, dw_ref) <= 1e-6
    ops.set_precision("bf16")


def test_fused_discriminator_block_refuses_second_derivative_without_the_branch():
    """A block that kept only sign bits cannot be differentiated twice: it says so instead of returning a wrong penalty."""
    from xmc_gan.model.df_gan import resD
    ops.set_precision("bf16")
    blk = resD(32, 64, downsample=True).to(DEV)
    x = torch.randn(2, 16, 16, 32, device=DEV).to(ops.act_dtype()).requires_grad_()
    y = blk(x)
    with pytest.raises(RuntimeError, match="second_order"):
        torch.autograd.grad(y.float().sum(), x, create_graph=True)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


@pytest.mark.parametrize("R,C", [(32, 27), (1, 1024), (64, 6912), (512, 8192), (256, 4096), (37, 53), (128, 128)])
@pytest.mark.parametrize("training", [True, False])
def test_spectral_weight_matches_legacy_hook_arithmetic(R, C, training):
    """ops.spectral_weight vs the arithmetic of torch.nn.utils.spectral_norm's compute_weight (reference
    model/modules.py:16-17,31-32) in f64 on the CPU: buffers after the power iteration, W/sigma, and dL/dW.
    Tolerance: f32 sums over <= 8192 terms -> 2e-5 relative L2."""
    g = torch.Generator().manual_seed(R * 131 + C)
    W = torch.randn(R, C, generator=g) * (2.0 / C) ** 0.5
    u0, v0 = torch.randn(R, generator=g), torch.randn(C, generator=g)
    G = torch.randn(R, C, generator=g)
    Wd = W.double().requires_grad_(True)
    u, v = u0.double(), v0.double()
    if training:
        v = torch.nn.functional.normalize(Wd.detach().t() @ u, dim=0, eps=1e-12)
        u = torch.nn.functional.normalize(Wd.detach() @ v, dim=0, eps=1e-12)
    sigma = torch.dot(u, Wd @ v)
    We = Wd / sigma
    (We * G.double()).sum().backward()

    Wp = W.to(DEV).requires_grad_(True)
    up, vp = u0.to(DEV).clone(), v0.to(DEV).clone()
    shape4 = (R, C, 1, 1) if C % 9 else (R, C // 9, 3, 3)
    Wp4 = Wp.view(shape4)
    Wep = ops.spectral_weight(Wp4, up, vp, training)
    assert Wep.shape == Wp4.shape
    (Wep * G.to(DEV).view(shape4)).sum().backward()
    rel = lambda a, b: ((a.double().cpu().flatten() - b.flatten()).norm() / b.norm().clamp_min(1e-30)).item()
    assert rel(Wep, We.detach()) < 2e-5
    assert rel(Wp.grad, Wd.grad) < 2e-5
    if training:
        assert rel(up, u) < 2e-5 and rel(vp, v) < 2e-5
    else:
        assert torch.equal(up.cpu(), u0) and torch.equal(vp.cpu(), v0)


@pytest.mark.parametrize("B,K,Hd,Cs,c_grad", [(256, 256, 256, [256, 256, 128, 64, 32, 32], False),
                                                (7, 20, 24, [5, 64, 65, 130], True),
                                                (64, 256, 256, [32] * 40, True)])
def test_cond_mlp_bank_matches_per_layer_reference(B, K, Hd, Cs, c_grad):
    """grouped-GEMM bank of conditioning MLPs (df_gan.py:232-241) vs Linear-ReLU-Linear in f64 on the CPU: outputs and
    every parameter gradient (and dc), including sizes that are not multiples of the 64x64 tile / 16-wide K chunk and more
    problems than one kernel-argument chunk (40 > 32).  f32 FMA sums over <= 256 terms: 1e-5 relative L2."""
    g = torch.Generator().manual_seed(B + K + len(Cs))
    c = torch.randn(B, K, generator=g)
    mk = lambda *s: torch.randn(*s, generator=g) * 0.2
    P = [(mk(Hd, K), mk(Hd), mk(C, Hd), mk(C)) for C in Cs]
    dY = [torch.randn(B, C, generator=g) for C in Cs]
    # reference
    cd = c.double().requires_grad_(c_grad)
    Pd = [tuple(t.double().requires_grad_(True) for t in m) for m in P]
    loss = 0
    ys_ref = []
    for (w1, b1, w2, b2), dy in zip(Pd, dY):
        y = torch.relu(cd @ w1.t() + b1) @ w2.t() + b2
        ys_ref.append(y.detach())
        loss = loss + (y * dy.double()).sum()
    loss.backward()
    # product
    cp = c.to(DEV).requires_grad_(c_grad)
    Pp = [tuple(t.to(DEV).requires_grad_(True) for t in m) for m in P]
    ys = ops.cond_mlp_bank(cp, Pp)
    assert len(ys) == len(Cs)
    sum((y * dy.to(DEV)).sum() for y, dy in zip(ys, dY)).backward()
    rel = lambda a, b: ((a.detach().double().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()
    for y, yr in zip(ys, ys_ref):
        assert y.shape == yr.shape and y.is_contiguous() and rel(y, yr) < 1e-5
    for mp, md in zip(Pp, Pd):
        for tp, td in zip(mp, md):
            assert tp.grad.shape == td.grad.shape and tp.grad.is_contiguous()
            assert rel(tp.grad, td.grad) < 1e-5
    if c_grad:
        assert rel(cp.grad, cd.grad) < 1e-5
    else:
        assert cp.grad is None


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("N,H,W", [(3, 4, 4), (2, 10, 10), (5, 64, 64), (64, 32, 32), (1, 128, 128)])
def test_region_attention_pooling(N, H, W, mode):
    """ops.attn_pool (CondConceptSampler / ConceptSampler, df_concept_gan.py:293-299, 570-578): per (sample, concept) softmax
    over H*W of scale * <q, key>, attention-weighted sum of x -- forward and the gradients w.r.t. key, q and x against an f64
    evaluation on the CPU.  Sizes cover one run per image, ragged runs (100 pixels), several runs per image and the
    many-images regime; the kernel recomputes the attention weights in backward instead of storing them."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(N * 1000 + H)
    ncon, pk, px, scale = 16, 4, 8, 0.5
    key = rt(torch.randn(N, H, W, ncon * pk, generator=g), mode)
    x = rt(torch.randn(N, H, W, ncon * px, generator=g), mode)
    q = torch.randn(N, ncon, pk, generator=g) * 2
    R = torch.randn(N, ncon, px, generator=g)
    kd, xd, qd = key.double().requires_grad_(), x.double().requires_grad_(), q.double().requires_grad_()
    sc = scale * torch.einsum("nck,npck->ncp", qd, kd.view(N, H * W, ncon, pk))
    a = torch.softmax(sc, dim=2)
    ctx_ref = torch.einsum("ncp,npcj->ncj", a, xd.view(N, H * W, ncon, px))
    (ctx_ref * R.double()).sum().backward()
    kp, xp = key.to(DEV, dt).requires_grad_(), x.to(DEV, dt).requires_grad_()
    qp = q.to(DEV).requires_grad_()
    ctx = ops.attn_pool(kp, qp, xp, ncon, scale)
    assert ctx.shape == (N, ncon, px) and ctx.dtype == torch.float32
    (ctx * R.to(DEV)).sum().backward()
    rel = lambda u, v: ((u.detach().double().cpu() - v).norm() / v.norm().clamp_min(1e-30)).item()
    t = 1e-5 if mode == "fp32" else 1e-2          # bf16: the gradients are stored in bf16
    assert rel(ctx, ctx_ref.detach()) < (1e-5 if mode == "fp32" else 1e-5), rel(ctx, ctx_ref.detach())
    assert rel(kp.grad, kd.grad) < t and rel(xp.grad, xd.grad) < t and rel(qp.grad, qd.grad) < 1e-4 + t


@pytest.mark.parametrize("mode", MODES + ["f16"])
@pytest.mark.parametrize("N,H,W,T", [(3, 16, 16, 18), (2, 10, 7, 5), (16, 32, 32, 18), (1, 128, 128, 24), (2, 8, 8, 32), (2, 4, 4, 16)])
def test_word_region_attention_pooling(N, H, W, T, mode):
    """ops.word_region_pool (the repaired concept_gan.InNetG's CondConceptSampler.get_context_embs, concept_gan.py:532-555): per
    (sample, concept, REGION) cosine scores against the word keys, padded words at -inf, softmax over the words, attention-weighted key
    sum, mean over regions -- forward and the gradients w.r.t. the query map and the keys against an f64 evaluation of the reference's
    formula on the CPU.  Sizes: one run per image, ragged runs (70 regions), several runs per image, the three compile-time word
    capacities (16 / 24 / 32), captions padded to different lengths."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(N * 1000 + H + T)
    ncon, pk = 16, 4
    qmap = rt(torch.randn(N, H, W, ncon * pk, generator=g) * 1.5, mode)
    kh = torch.nn.functional.normalize(torch.randn(N, ncon, T, pk, generator=g), dim=3)
    lens = torch.randint(1, T + 1, (N,), generator=g)
    lens[0] = T
    pad = torch.arange(T)[None, :] >= lens[:, None]
    R = torch.randn(N, ncon, pk, generator=g)
    qd, kd = qmap.double().requires_grad_(), kh.double().requires_grad_()
    # the reference's formula: img [N,C,p',HW] and words [N,C,p',T] normalised over p', sim [N,C,HW,T], masked softmax over T
    qn = torch.nn.functional.normalize(qd.view(N, H * W, ncon, pk).permute(0, 2, 3, 1), p=2, dim=2)
    kn = kd.permute(0, 1, 3, 2)
    sim = torch.matmul(qn.transpose(2, 3), kn).masked_fill(pad.view(N, 1, 1, T), float("-inf"))
    ctx_ref = torch.matmul(torch.softmax(sim, dim=3), kn.transpose(2, 3)).mean(dim=2)
    (ctx_ref * R.double()).sum().backward()
    qp, kp = qmap.to(DEV, dt).requires_grad_(), kh.to(DEV).requires_grad_()
    ctx = ops.word_region_pool(qp, kp, pad.to(DEV))
    assert ctx.shape == (N, ncon, pk) and ctx.dtype == torch.float32
    (ctx * R.to(DEV)).sum().backward()
    rel = lambda u, v: ((u.detach().double().cpu() - v).norm() / v.norm().clamp_min(1e-30)).item()
    assert rel(ctx, ctx_ref.detach()) < 1e-5, rel(ctx, ctx_ref.detach())
    assert rel(kp.grad, kd.grad) < 1e-4, rel(kp.grad, kd.grad)
    assert (kp.grad.cpu()[pad[:, None, :, None].expand_as(kh)] == 0).all()          # a padding word's key receives nothing
    # the query gradient is stored in the activation format: 1e-2 in the 16-bit modes
    assert rel(qp.grad, qd.grad) < (1e-5 if mode == "fp32" else 1e-2), rel(qp.grad, qd.grad)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("N,H,W,C", [(3, 16, 16, 128), (2, 100, 7, 64), (64, 32, 32, 128), (1, 128, 128, 128), (2, 20, 20, 8)])
def test_global_avgpool_large_maps(N, H, W, C, mode):
    """F.adaptive_avg_pool2d(x, 1) on the maps the concept samplers pool (df_concept_gan.py:557): the many-workgroup kernel
    (H*W >= 256) against the CPU, forward and backward."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(H * W + C)
    x = rt(torch.randn(N, H, W, C, generator=g) + 0.3, mode)
    r = torch.randn(N, C, generator=g)
    xd = x.to(DEV, dt).requires_grad_()
    y = ops.global_avgpool(xd)
    assert y.shape == (N, C) and y.dtype == torch.float32
    torch.testing.assert_close(y.cpu(), x.double().mean(dim=(1, 2)).float(), rtol=1e-5, atol=1e-5)
    (y * r.to(DEV)).sum().backward()
    ref = (r / (H * W))[:, None, None, :].expand(N, H, W, C)
    torch.testing.assert_close(xd.grad.float().cpu(), rt(ref, mode), rtol=1e-2 if mode == "bf16" else 1e-6, atol=1e-8)


def test_grad_penalty_matches_reference_expression():
    """ops.grad_penalty == mean(sqrt(sum(cat(g0, g1)**2, 1))**6) (train_gan.py:241-247) and its gradient, f64 reference;
    block widths that are and are not multiples of 4, a 3*64*64-wide image block."""
    g = torch.Generator().manual_seed(3)
    for B, shapes in ((5, [(3, 8, 8), (6,)]), (4, [(3, 64, 64), (256,)]), (3, [(10,), (7,)])):
        blocks = [torch.randn(B, *s, generator=g) * 0.3 for s in shapes]
        ref_in = [b.double().requires_grad_() for b in blocks]
        ref = (torch.cat([b.reshape(B, -1) for b in ref_in], 1).pow(2).sum(1).sqrt() ** 6).mean()
        ref.backward()
        dev = [b.to(DEV).requires_grad_() for b in blocks]
        gp = ops.grad_penalty(*dev)
        (2.0 * gp).backward()
        assert abs(gp.item() - ref.item()) <= 1e-5 * abs(ref.item())
        for d_, r_ in zip(dev, ref_in):
            torch.testing.assert_close(d_.grad.cpu().double(), 2.0 * r_.grad, rtol=1e-5, atol=1e-7 * float(r_.grad.abs().max()))


def test_concept_algebra_kernels_match_composed_reference():
    """xmc_concept_query / xmc_concept_head (csrc/concept.hip) against the composed f64 expressions of the reference
    (df_concept_gan.py:238-253, 273-326): outputs and every gradient (inputs and all parameters)."""
    g = torch.Generator().manual_seed(5)
    B, E = 5, 256
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    sent = rnd(B, E)
    wq, gnw, gnb = rnd(64, E, 1, 1, sc=E ** -0.5), 1 + 0.1 * rnd(64), 0.1 * rnd(64)
    pooled = rnd(B, 16, 8)
    P = [rnd(64, 8, 1, 1, sc=0.4), rnd(16, 4, sc=0.5)]
    for _ in range(2):
        P += [rnd(128, E + 4, 1, 1, sc=(E + 4) ** -0.5), 0.1 * rnd(128), rnd(128, 8, 1, 1, sc=0.35), 0.1 * rnd(128)]

    def ref(sent, wq, gnw, gnb, pooled, P, norm):
        q = torch.einsum('bi,goi->bgo', sent, wq.view(16, 4, E))
        if norm:
            q = F.group_norm(q.reshape(B, 64), 16, gnw, gnb).view(B, 16, 4)
        v = torch.einsum('bgi,goi->bgo', pooled, P[0].view(16, 4, 8))
        adj = torch.tanh(F.linear(v, P[1]))
        r = F.relu(v + torch.matmul(adj, v))
        cond = torch.cat([sent.view(B, 1, E).expand(B, 16, E), r], dim=2)
        outs = []
        for t in range(2):
            w1, b1, w2, b2 = P[2 + 4 * t: 6 + 4 * t]
            h = F.leaky_relu(torch.einsum('bgi,goi->bgo', cond, w1.view(16, 8, E + 4)) + b1.view(1, 16, 8), 0.2)
            outs.append((torch.einsum('bgi,goi->bgo', h, w2.view(16, 8, 8)) + b2.view(1, 16, 8)).reshape(B, 128))
        return q, outs[0], outs[1]

    for norm in (True, False):
        ins64 = [t.double().requires_grad_() for t in (sent, wq, gnw, gnb, pooled)] + [[p.double().requires_grad_() for p in P]]
        q_r, ga_r, be_r = ref(*ins64, norm)
        wts = [rnd(*t.shape).double() for t in (q_r, ga_r, be_r)]
        (q_r * wts[0]).sum().add((ga_r * wts[1]).sum()).add((be_r * wts[2]).sum()).backward()
        d = lambda t: t.to(DEV).requires_grad_()
        sent_d, wq_d, gnw_d, gnb_d, pooled_d = d(sent), d(wq), d(gnw), d(gnb), d(pooled)
        P_d = [d(p) for p in P]
        q = ops.concept_query(sent_d, wq_d, gnw_d if norm else None, gnb_d if norm else None)
        ga, be = ops.concept_head(pooled_d, sent_d, P_d)
        ((q * wts[0].float().to(DEV)).sum() + (ga * wts[1].float().to(DEV)).sum() + (be * wts[2].float().to(DEV)).sum()).backward()
        tol_ = dict(rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(q.detach().cpu().double(), q_r.detach(), **tol_)
        torch.testing.assert_close(ga.detach().cpu().double(), ga_r.detach(), **tol_)
        torch.testing.assert_close(be.detach().cpu().double(), be_r.detach(), **tol_)
        pairs = [(sent_d, ins64[0]), (wq_d, ins64[1]), (pooled_d, ins64[4])] + list(zip(P_d, ins64[5]))
        if norm:
            pairs += [(gnw_d, ins64[2]), (gnb_d, ins64[3])]
        for got, want in pairs:
            sc_ = float(want.grad.abs().max())
            torch.testing.assert_close(got.grad.cpu().double(), want.grad, rtol=5e-4, atol=5e-5 * max(sc_, 1e-3))


@pytest.mark.parametrize("E", [256, 768])
def test_self_attention_concept_algebra_matches_composed_reference(E):
    """xmc_concept_gquery / xmc_concept_head with sent_linear (the self-attention block, df_concept_gan.py:443-478, 555-581)
    against the composed f64 expressions: outputs and every gradient."""
    g = torch.Generator().manual_seed(9)
    B = 6
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    sent, q0, pooled = rnd(B, E), rnd(B, 128), rnd(B, 16, 8)
    wq, gnw, gnb = rnd(64, 8, 1, 1, sc=0.4), 1 + 0.1 * rnd(64), 0.1 * rnd(64)
    P = [rnd(64, 8, 1, 1, sc=0.4), rnd(16, 4, sc=0.5)]
    for _ in range(2):
        P += [rnd(128, E + 4, 1, 1, sc=(E + 4) ** -0.5), 0.1 * rnd(128), rnd(128, 8, 1, 1, sc=0.35), 0.1 * rnd(128)]
    P += [rnd(4, E, sc=2.0 * E ** -0.5)]                      # sent_linear

    def ref(sent, q0, wq, gnw, gnb, pooled, P):
        q = torch.einsum('bgi,goi->bgo', q0.view(B, 16, 8), wq.view(16, 4, 8))
        q = F.group_norm(q.reshape(B, 64), 16, gnw, gnb).view(B, 16, 4)
        v = torch.einsum('bgi,goi->bgo', pooled, P[0].view(16, 4, 8))
        adj = torch.tanh(F.linear(v, P[1]))
        st = F.relu(v + torch.matmul(adj, v)).transpose(1, 2)               # [B,4,16]
        s = F.linear(sent, P[10]).view(B, -1, 1)
        attn = F.softmax(torch.matmul(s.transpose(1, 2), st), dim=2)
        ctx = (st * attn).transpose(1, 2)
        cond = torch.cat([sent.view(B, 1, E).expand(B, 16, E), ctx], dim=2)
        outs = []
        for t in range(2):
            w1, b1, w2, b2 = P[2 + 4 * t: 6 + 4 * t]
            h = F.leaky_relu(torch.einsum('bgi,goi->bgo', cond, w1.view(16, 8, E + 4)) + b1.view(1, 16, 8), 0.2)
            outs.append((torch.einsum('bgi,goi->bgo', h, w2.view(16, 8, 8)) + b2.view(1, 16, 8)).reshape(B, 128))
        return q, outs[0], outs[1]

    ins64 = [t.double().requires_grad_() for t in (sent, q0, wq, gnw, gnb, pooled)] + [[p.double().requires_grad_() for p in P]]
    q_r, ga_r, be_r = ref(*ins64)
    wts = [rnd(*t.shape).double() for t in (q_r, ga_r, be_r)]
    (q_r * wts[0]).sum().add((ga_r * wts[1]).sum()).add((be_r * wts[2]).sum()).backward()
    d = lambda t: t.to(DEV).requires_grad_()
    sent_d, q0_d, wq_d, gnw_d, gnb_d, pooled_d = d(sent), d(q0), d(wq), d(gnw), d(gnb), d(pooled)
    P_d = [d(p) for p in P]
    q = ops.concept_gquery(q0_d, wq_d, gnw_d, gnb_d)
    ga, be = ops.concept_head(pooled_d, sent_d, P_d)
    ((q * wts[0].float().to(DEV)).sum() + (ga * wts[1].float().to(DEV)).sum() + (be * wts[2].float().to(DEV)).sum()).backward()
    tol_ = dict(rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(q.detach().cpu().double(), q_r.detach(), **tol_)
    torch.testing.assert_close(ga.detach().cpu().double(), ga_r.detach(), **tol_)
    torch.testing.assert_close(be.detach().cpu().double(), be_r.detach(), **tol_)
    pairs = [(sent_d, ins64[0]), (q0_d, ins64[1]), (wq_d, ins64[2]), (gnw_d, ins64[3]), (gnb_d, ins64[4]), (pooled_d, ins64[5])]
    pairs += list(zip(P_d, ins64[6]))
    for got, want in pairs:
        sc_ = float(want.grad.abs().max())
        torch.testing.assert_close(got.grad.cpu().double(), want.grad, rtol=5e-4, atol=5e-5 * max(sc_, 1e-3))


def test_optimizer_step_repacks_cached_weights_in_bulk():
    """HipAdam.step changes the parameters behind autograd's back; every cached packed copy (forward, data-gradient, fused
    upsample, grouped) must be re-packed by the step itself -- same buffers, marked valid -- and give the new weights' result."""
    from xmc_gan_amd.optim import HipAdam
    ops.set_precision("bf16")
    g = torch.Generator().manual_seed(5)
    geoms = [ops.ConvGeom(32, 64, 3, 1, 1), ops.ConvGeom(64, 32, 3, 1, 1), ops.ConvGeom(128, 128, 3, 1, 1, groups=16)]
    ws = [torch.nn.Parameter((torch.randn(gm.cout, gm.cin // gm.groups, 3, 3, generator=g) * 0.1).to(DEV)) for gm in geoms]
    xs = [torch.randn(2, 16, 16, gm.cin, generator=g).to(DEV, torch.bfloat16).requires_grad_() for gm in geoms]
    opt = HipAdam(ws, lr=0.05)

    def run():
        ys = [ops.conv2d(xs[0], ws[0], None, geoms[0]), ops.upconv3x3(xs[1], ws[1], None, geoms[1]),
              ops.conv2d(xs[2], ws[2], None, geoms[2])]
        sum((y.float() ** 2).sum() for y in ys).backward()
        return [y.detach().float() for y in ys]

    run()                                                    # creates the forward and data-gradient entries
    ents = {k: e for k, e in ops._pack_cache.items() if any(e.ref() is w for w in ws)}
    assert len(ents) >= 6
    ptrs = {k: e.out.data_ptr() for k, e in ents.items()}
    before = [w.detach().clone() for w in ws]
    opt.step()
    assert all(not torch.equal(b, w.detach()) for b, w in zip(before, ws))
    for k, e in ents.items():
        w = e.ref()
        assert ops._pack_cache[k] is e and e.out.data_ptr() == ptrs[k] and e.valid(w, e.geom), k
    n_before = len(ops._pack_cache)
    ys = run()
    assert len(ops._pack_cache) == n_before and all(ops._pack_cache[k] is e for k, e in ents.items())     # no lazy re-pack happened
    # ... and the buffers really hold the new weights: same result as a freshly packed non-Parameter copy
    y_ref = ops.conv2d(xs[0].detach(), ws[0].detach().clone(), None, geoms[0]).float()
    assert torch.equal(ys[0], y_ref)
    y_ref = ops.conv2d(xs[2].detach(), ws[2].detach().clone(), None, geoms[2]).float()
    assert torch.equal(ys[2], y_ref)


@pytest.mark.parametrize("mode", MODES)
def test_affine_skip_node_sums_both_gradients_of_its_input(mode):
    """ops.affine2_lrelu_skip returns (h, x): the gradient arriving at the second output is added inside the affine backward
    kernel.  Reference: the plain affine node plus autograd's own sum."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(11)
    N, H, C = 3, 16, 64
    x0 = torch.randn(N, H, H, C, generator=g).to(DEV, dt)
    mods = [torch.randn(N, C, generator=g).to(DEV) * 0.5 + (1.0 if i % 2 == 0 else 0.0) for i in range(4)]
    r1 = torch.randn(N, H, H, C, generator=g).to(DEV, dt)
    r2 = torch.randn(N, H, H, C, generator=g).to(DEV, dt)
    outs = []
    for skip in (True, False):
        x = x0.clone().requires_grad_()
        ms = [m.clone().requires_grad_() for m in mods]
        if skip:
            h, xs = ops.affine2_lrelu_skip(x, *ms)
        else:
            h, xs = ops.affine2_lrelu(x, *ms), x
        ((h.float() * r1.float()).sum() + (xs.float() * r2.float()).sum()).backward()
        outs.append([h.detach().float(), x.grad.float()] + [m.grad for m in ms])
    for a, b in zip(*outs):
        sc = b.abs().max().item() + 1e-12
        torch.testing.assert_close(a, b, rtol=2e-2 if mode == "bf16" else 1e-5, atol=(2e-2 if mode == "bf16" else 1e-5) * sc)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("norm,with_sl", [(True, False), (False, False), (True, True)])
def test_concept_stage_node_equals_composed_operators(mode, norm, with_sl):
    """ops.concept_stage (key projection [+ GroupNorm], region attention, concept head, modulation as ONE autograd node with the
    three gradients of its input summed inside the kernels) against the same stage composed from the separate operators."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(3 + int(norm) + 2 * int(with_sl))
    B, H, E = 3, 16, 256
    rnd = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(DEV)
    x0 = rnd(B, H, H, 128).to(dt)
    q0, sent0 = rnd(B, 16, 4), rnd(B, E)
    wk0 = rnd(64, 8, 1, 1, sc=0.35)
    gnw0, gnb0 = 1 + 0.1 * rnd(64), 0.1 * rnd(64)
    P0 = [rnd(64, 8, 1, 1, sc=0.4), rnd(16, 4, sc=0.5)]
    for _ in range(2):
        P0 += [rnd(128, E + 4, 1, 1, sc=(E + 4) ** -0.5), 0.1 * rnd(128), rnd(128, 8, 1, 1, sc=0.35), 0.1 * rnd(128)]
    if with_sl:
        P0 += [rnd(4, E, sc=2.0 * E ** -0.5)]
    r = rnd(B, H, H, 128).to(dt)
    geom = ops.ConvGeom(128, 64, 1, 1, 0, groups=16)
    res = []
    for fused in (True, False):
        leaf = lambda t: t.clone().requires_grad_()
        x, q, sent, wk, gnw, gnb = leaf(x0), leaf(q0), leaf(sent0), torch.nn.Parameter(wk0.clone()), leaf(gnw0), leaf(gnb0)
        P = [leaf(p) for p in P0]
        if fused:
            y = ops.concept_stage(x, q, sent, wk, gnw if norm else None, gnb if norm else None, geom, 16, 0.7, P)
        else:
            key = ops.conv2d(x, wk, None, geom)
            if norm:
                key = ops.groupnorm(key, gnw, gnb, 16)
            pooled = ops.attn_pool(key, q, x, 16, 0.7)
            gamma, beta = ops.concept_head(pooled, sent, P)
            y = ops.affine_lrelu(x, gamma, beta)
        (y.float() * r.float()).sum().backward()
        res.append([y.detach().float(), x.grad.float(), q.grad, sent.grad, wk.grad] + ([gnw.grad, gnb.grad] if norm else []) + [p.grad for p in P])
    # (the GroupNorm bias shifts every score of a concept by the same amount and the softmax over the pixels ignores it: its gradient is
    # zero in exact arithmetic and 1e-8-sized summation-order noise in both forms -- it is held to the scale of its sibling, the GroupNorm
    # weight's gradient, not to its own)
    floor = 1e-3 * max(b.abs().max().item() for b in res[1][4:])
    for k, (a, b) in enumerate(zip(*res)):
        sc = max(b.abs().max().item(), floor if k >= 4 else 0.0) + 1e-12
        # bf16: the composed form rounds each of the three gradients of x to bf16 before adding them, the node adds in f32
        torch.testing.assert_close(a, b, rtol=3e-2 if mode == "bf16" else 2e-4, atol=(3e-2 if mode == "bf16" else 2e-4) * sc, msg=lambda m: f"tensor {k}: {m}")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("C,H,tail,gamma", [(32, 32, True, 0.3), (32, 32, True, 0.0), (64, 16, False, 0.3), (128, 16, False, 0.0),
                                            (256, 8, False, -0.5)])
def test_generator_block_end_node_equals_composed_operators(mode, C, H, tail, gamma):
    """ops.g_block_end (affine pair, c2, block sum with the upsampled shortcut [, LeakyReLU, conv_out, tanh] as ONE node whose
    backward takes d(gamma) from <c2^T dout, h2> + <b2, colsum(dout)> instead of a stored c2 output, and the LeakyReLU' mask of
    the tail inside conv_out's data gradient) against the same operators composed.  gamma = 0 is the reference's initial value
    (df_gan.py:196): d(gamma) must survive it.  Followed by an affine-skip node of a next block, whose backward hands the 2x2 sum
    pool of its dx to this node's backward (ops._pooled_grads) -- both ways must agree with the pooling pass."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(7 + C + H)
    N = 3
    rnd = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(DEV)
    h10 = rnd(N, H, H, C).to(dt)
    sc0 = rnd(N, H // 2, H // 2, C).to(dt)
    mods0 = [rnd(N, C, sc=0.5) + (1.0 if i % 2 == 0 else 0.0) for i in range(4)]
    mods_next0 = [rnd(N, C, sc=0.5) + (1.0 if i % 2 == 0 else 0.0) for i in range(4)]
    w20, b20 = rnd(C, C, 3, 3, sc=(9 * C) ** -0.5), rnd(C, sc=0.1)
    wo0, bo0 = rnd(3, C, 3, 3, sc=(9 * C) ** -0.5), rnd(3, sc=0.1)
    geom2, geomo = ops.ConvGeom(C, C, 3, 1, 1), ops.ConvGeom(C, 3, 3, 1, 1)
    r_img = rnd(N, H, H, 8).to(dt)
    r_h, r_x = rnd(N, H, H, C).to(dt), rnd(N, H, H, C).to(dt)
    res = []
    # fp32: fused against composed.  Half formats: both against the composed operators in fp32 -- the fused node rounds FEWER
    # intermediates (gamma * dout and the masked gradient are never stored), so the bar is "no further from the f32 result than
    # the composed half-precision operators are" (x2 + a floor), not agreement between two differently rounded half results.
    variants = [(True, mode), (False, mode)] + ([(False, "fp32")] if mode != "fp32" else [])
    for fused, vmode in variants:
        ops.set_precision(vmode)
        vdt = ops.act_dtype()
        leaf = lambda t: t.float().to(vdt).clone().requires_grad_() if t.dtype != torch.float32 else t.clone().requires_grad_()
        h1, sc = leaf(h10), leaf(sc0)
        ms, msn = [leaf(m) for m in mods0], [leaf(m) for m in mods_next0]
        w2, b2, wo, bo = (torch.nn.Parameter(t.clone()) for t in (w20, b20, wo0, bo0))
        gam = torch.nn.Parameter(torch.full((1,), gamma, device=DEV))
        if fused:
            y = ops.g_block_end(h1, ms, w2, b2, geom2, sc, gam, tail=(wo, bo, geomo) if tail else None)
        else:
            h2 = ops.affine2_lrelu(h1, *ms)
            y = ops.axpby_up(sc, ops.conv2d(h2, w2, b2, geom2), gam, lrelu=tail)
            if tail:
                y = ops.conv2d(y, wo, bo, geomo, act=L.ACT_TANH)
        if tail:
            loss = (y.float() * r_img.float()).sum()
        else:       # the next block's first node: its dx is this node's dout, its pooled dx this node's shortcut gradient
            hn, xs = ops.affine2_lrelu_skip(y, *msn, pool_grad=fused)
            loss = (hn.float() * r_h.float()).sum() + (xs.float() * r_x.float()).sum()
        ops.new_iteration(DEV)
        loss.backward()
        assert not ops._pooled_grads, "the pooled by-product was not consumed"
        res.append([y.detach().float(), h1.grad.float(), sc.grad.float(), gam.grad, w2.grad, b2.grad] + [m.grad for m in ms] +
                   ([wo.grad, bo.grad] if tail else [m.grad for m in msn]))
    ops.set_precision(mode)
    names = ["y", "dh1", "dsc", "dgamma", "dw2", "db2", "dg0", "db0", "dg1", "db1"] + (["dwo", "dbo"] if tail else ["n0", "n1", "n2", "n3"])
    for k, nm in enumerate(names):
        fz, cp = res[0][k], res[1][k]
        truth = res[2][k] if mode != "fp32" else cp
        scl = truth.abs().max().item() + 1e-12
        ef = (fz - truth).abs().max().item() / scl
        if mode == "fp32":
            assert ef < 2e-4, (nm, ef)
        else:
            ec = (cp - truth).abs().max().item() / scl
            assert ef <= 2 * ec + 1e-2, (nm, ef, ec)


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("N,H,W", [(2, 64, 64), (3, 32, 128), (1, 256, 256)])
def test_discriminator_stem_composition_kernels(N, H, W, mode):
    """csrc/dstem.hip: conv_img -> conv_r[0] (4x4 stride 2) and conv_img -> avg_pool2d -> conv_s as one 6x6 stride-2 convolution of
    the image with composed weights (ops.compose_dstem), against the two-stage computation in f32 on the CPU (df_gan.py:114,127,
    272-291): interior pixels of h1 and every pixel of the shortcut must agree, and after the border kernel every pixel of h1; the
    weight-gradient kernel against autograd through the same composed convolution (the gradients of the border tables are checked
    end to end by test_composed_stem_block_equals_conv_img_plus_block)."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(N + H)
    x = rt(torch.rand(N, 3, H, W, generator=g) * 2 - 1, mode)
    w_img, b_img = torch.randn(32, 3, 3, 3, generator=g) / math.sqrt(27), torch.randn(32, generator=g) * 0.1
    w0 = torch.randn(64, 32, 4, 4, generator=g) / math.sqrt(512)
    ws, bs = torch.randn(64, 32, 1, 1, generator=g) / math.sqrt(32), torch.randn(64, generator=g) * 0.1
    ci = F.conv2d(x, w_img, b_img, 1, 1)
    h1_ref = F.leaky_relu(F.conv2d(ci, w0, None, 2, 1), 0.2)
    sc_ref = F.conv2d(F.avg_pool2d(ci, 2), ws, bs)
    wsets, bias, D, DB = ops.compose_dstem(w_img.to(DEV), b_img.to(DEV), w0.to(DEV), ws.to(DEV), bs.to(DEV))
    # the composition itself, in f32 on the CPU: a 6x6 stride-2 pad-2 convolution
    wc = wsets.cpu().view(128, 6, 6, 8).permute(0, 3, 1, 2)[:, :3]
    yc = F.conv2d(x, wc, bias.cpu(), 2, 2)
    assert (F.leaky_relu(yc[:, :64], 0.2) - h1_ref)[:, :, 1:-1, 1:-1].abs().max() < 1e-4 and (yc[:, 64:] - sc_ref).abs().max() < 1e-4
    xin = to_nhwc(x, 8, dt)
    h1, sc = ops._dstem_fwd_raw(xin, wsets, bias)
    assert L.load().xmc_last_kernel().decode() == "dstem_fwd_kernel"
    t = dict(rtol=2e-2, atol=2e-2 * h1_ref.abs().max().item()) if mode == "bf16" else dict(rtol=3e-3, atol=3e-3 * h1_ref.abs().max().item())
    torch.testing.assert_close(from_nhwc(h1, 64)[:, :, 1:-1, 1:-1], h1_ref[:, :, 1:-1, 1:-1], **t)
    torch.testing.assert_close(from_nhwc(sc, 64), sc_ref, **t)
    # ... and with the border corrections (conv_r[0]'s zero padding of conv_img's output) every pixel of h1
    ops._dstem_border_fwd_raw(xin, wsets, bias, D, DB, h1)
    torch.testing.assert_close(from_nhwc(h1, 64), h1_ref, **t)
    # weight gradient: d/dW of sum(y * r) for y = conv6x6(x; W) = sum_pixels r (x) patch; border pixels of the h1 half excluded
    r = rt(torch.randn(N, 128, H // 2, W // 2, generator=g), mode)
    rm = r.clone()
    rm[:, :64, 0] = 0; rm[:, :64, -1] = 0; rm[:, :64, :, 0] = 0; rm[:, :64, :, -1] = 0
    wc8 = torch.zeros(128, 8, 6, 6, requires_grad=True)
    bc = torch.zeros(128, requires_grad=True)
    x8 = torch.cat((x, torch.zeros(N, 5, H, W)), 1)
    (F.conv2d(x8, wc8, bc, 2, 2) * rm).sum().backward()
    ops.new_iteration(DEV)
    dw, db, _, _ = ops._dstem_wgrad_raw(xin, to_nhwc(r[:, :64], 64, dt), to_nhwc(r[:, 64:], 64, dt), skip_border=True, border=False)
    ref = wc8.grad.permute(0, 2, 3, 1).reshape(128, 36, 8)
    assert ((dw.cpu() - ref).norm() / ref.norm()) < 5e-5, (dw.cpu() - ref).norm() / ref.norm()
    assert ((db.cpu() - bc.grad).norm() / bc.grad.norm()) < 5e-5
    # gradient of the image: the adjoint of the composed stem incl. its border corrections == autograd through the two-stage form
    xr = x.clone().requires_grad_()
    cir = F.conv2d(xr, w_img, b_img, 1, 1)
    ((F.conv2d(cir, w0, None, 2, 1) * r[:, :64]).sum() + (F.conv2d(F.avg_pool2d(cir, 2), ws, bs) * r[:, 64:]).sum()).backward()
    dimg = ops._dstem_dgrad_raw(to_nhwc(r[:, :64], 64, dt), to_nhwc(r[:, 64:], 64, dt), wsets, D, H, W)
    assert L.load().xmc_last_kernel().decode() == "dstem_dgrad_kernel"
    got = from_nhwc(dimg, 3)
    e_all, e_in = rel_l2(got, xr.grad), rel_l2(got[:, :, 2:-2, 2:-2], xr.grad[:, :, 2:-2, 2:-2])
    e_b = rel_l2(torch.cat((got[:, :, 0], got[:, :, -1], got[:, :, :, 0], got[:, :, :, -1]), 2), torch.cat((xr.grad[:, :, 0], xr.grad[:, :, -1], xr.grad[:, :, :, 0], xr.grad[:, :, :, -1]), 2))
    assert max(e_all, e_in, e_b) < (1.5e-2 if mode == "bf16" else 3e-3), (e_all, e_in, e_b)
    assert float(dimg[..., 3:].abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("N,H,W,keep,pool", [(3, 64, 64, True, True), (2, 128, 192, True, True), (5, 64, 128, False, True), (2, 32, 64, True, False)])
def test_block_end_recomputes_the_stem_shortcut_from_the_image(N, H, W, keep, pool, mode):
    """XmcConvDesc.sc_img (xmc_conv_ptile_scimg): the first block's conv_r[2] + block sum with its residual -- the composed stem's
    shortcut -- recomputed per tile from the image (16 MFMAs per wave on an 18 x 66 pixel patch) against the same launch reading the
    shortcut tensor the stem kernel would have written: output, sign bytes and pooled output.  Also: the stem kernel asked for no
    shortcut writes the same h1."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    g = torch.Generator().manual_seed(N * 11 + H + W)
    x = rt(torch.rand(N, 3, H, W, generator=g) * 2 - 1, mode)
    w_img, b_img = torch.randn(32, 3, 3, 3, generator=g) / math.sqrt(27), torch.randn(32, generator=g) * 0.1
    w0 = torch.randn(64, 32, 4, 4, generator=g) / math.sqrt(512)
    ws, bs = torch.randn(64, 32, 1, 1, generator=g) / math.sqrt(32), torch.randn(64, generator=g) * 0.1
    w2 = (torch.randn(64, 64, 3, 3, generator=g) / math.sqrt(576)).to(DEV)
    wsets, bias, D, DB = ops.compose_dstem(w_img.to(DEV), b_img.to(DEV), w0.to(DEV), ws.to(DEV), bs.to(DEV))
    xin = to_nhwc(x, 8, dt)
    h1, sc = ops._dstem_fwd_raw(xin, wsets, bias)
    h1b, none = ops._dstem_fwd_raw(xin, wsets, bias, want_sc=False)
    assert none is None and torch.equal(h1, h1b)
    g2 = ops.ConvGeom(64, 64, 3, 1, 1)
    al = torch.tensor([0.59], device=DEV)
    ref = ops._conv_fwd_raw(h1, w2, None, g2, L.ACT_LRELU, dt, res=sc, alpha=al, want_sign=keep, want_pool=pool, round_act=True)
    got = ops._conv_fwd_raw(h1, w2, None, g2, L.ACT_LRELU, dt, alpha=al, want_sign=keep, want_pool=pool, round_act=True,
                            sc_img=ops._dstem_sc_operands(xin, wsets, bias))
    if not pool:                  # only the two block-end sets with a pooled output exist with a recomputed shortcut: declined, nothing launched
        assert got is None
        return
    assert got is not None and L.load().xmc_last_kernel().decode().endswith("sc>")
    # the shortcut's f32 sum is accumulated in another order (32x32x16 MFMAs over window rows, the stem kernel: 16x16x32 over tap pairs),
    # so its rounding to the 16-bit format differs by one ulp on a few elements in a thousand: sign bytes equal, values to 1e-4
    if keep:
        assert torch.equal(got[1], ref[1])
    for a, b in zip(got, ref):
        if a.dtype != torch.uint8:
            e = rel_l2(a, b)
            assert e <= (2e-3 if mode == "bf16" else 2e-4), e
            assert (a.float() - b.float()).abs().max() <= (2.0 ** -7 if mode == "bf16" else 2.0 ** -10) * b.float().abs().max()
    ops.set_precision("bf16")


@pytest.mark.parametrize("mode,N,S", [("f16", 3, 64), ("bf16", 2, 128), ("bf16", 8, 64)])
def test_composed_stem_block_equals_conv_img_plus_block(mode, N, S):
    """ops.DStemBlockFn (conv_img + the first discriminator block on the composed stem, border pixels on strips) against the form it
    replaces -- conv_img, then ops.ResDFn -- on the same parameters and image: block output, pooled output and the gradients of all
    seven parameter tensors under a random linear loss.  The two forms round different intermediates to 16 bits (the composed one
    fewer), so they are compared with each other at the format's tolerance and, border pixels included, with the f32 CPU reference."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    from xmc_gan.model.df_gan import resD
    from xmc_gan.model.modules import HipConv2d
    torch.manual_seed(5)
    conv_img = HipConv2d(3, 32, 3, 1, 1).to(DEV)
    blk = resD(32, 64, True).to(DEV)
    with torch.no_grad():
        blk.gamma.fill_(0.7)
        conv_img.bias.normal_(0, 0.1)
        blk.conv_s.bias.normal_(0, 0.1)
    g = torch.Generator().manual_seed(N + S)
    x = rt(torch.rand(N, 3, S, S, generator=g) * 2 - 1, mode)
    xin = to_nhwc(x, 8, dt).requires_grad_()
    r = rt(torch.randn(N, 64, S // 2, S // 2, generator=g), mode)
    params = [conv_img.weight, conv_img.bias, blk.conv_r[0].weight, blk.conv_r[2].weight, blk.conv_s.weight, blk.conv_s.bias, blk.gamma]

    def grads(out):
        for p_ in params + [xin]:
            p_.grad = None
        ops.new_iteration(DEV)
        (out.float() * to_nhwc(r, 64, torch.float32)).sum().backward()
        return [p_.grad.detach().float().cpu().clone() for p_ in params] + [from_nhwc(xin.grad, 3)]

    ci, cip = conv_img(xin, want_pool=True)
    out_o, pool_o = blk(ci, xp_hint=cip, want_pool=True)
    g_o = grads(out_o)
    r0, r2, s_ = blk.conv_r[0], blk.conv_r[2], blk.conv_s
    out_n, pool_n = ops.DStemBlockFn.apply(xin, conv_img.weight, conv_img.bias, r0.weight, r2.weight, s_.weight, s_.bias, blk.gamma,
                                           conv_img.geom, r0.geom, r2.geom, s_.geom, True)
    g_n = grads(out_n)
    # f32 reference on the CPU
    P = [p_.detach().float().cpu().clone().requires_grad_() for p_ in params] + [x.clone().requires_grad_()]
    ci_r = F.conv2d(P[7], P[0], P[1], 1, 1)
    br = F.leaky_relu(F.conv2d(F.leaky_relu(F.conv2d(ci_r, P[2], None, 2, 1), 0.2), P[3], None, 1, 1), 0.2)
    out_r = F.conv2d(F.avg_pool2d(ci_r, 2), P[4], P[5]) + P[6] * br
    (out_r * r).sum().backward()
    tl = 2e-2 if mode == "bf16" else 3e-3
    sc_ = out_r.abs().max().item()
    torch.testing.assert_close(from_nhwc(out_n, 64), out_r.detach(), rtol=tl, atol=tl * sc_)
    torch.testing.assert_close(from_nhwc(out_n, 64), from_nhwc(out_o, 64), rtol=tl, atol=tl * sc_)
    torch.testing.assert_close(from_nhwc(pool_n, 64), F.avg_pool2d(out_r.detach(), 2), rtol=tl, atol=tl * sc_)
    names = ["conv_img.weight", "conv_img.bias", "conv_r.0.weight", "conv_r.2.weight", "conv_s.weight", "conv_s.bias", "gamma", "image"]
    for n_, a, b, c in zip(names, g_n, g_o, P):
        e_ref, e_old = rel_l2(a, c.grad), rel_l2(b, c.grad)
        print(f"{n_:18s} composed vs f32 {e_ref:.2e}   conv_img + ResDFn vs f32 {e_old:.2e}")
        assert e_ref < (3e-2 if mode == "bf16" else 4e-3) or e_ref < 1.5 * e_old + 1e-3, (n_, e_ref, e_old)


@pytest.mark.parametrize("mode,N,S", [("f16", 3, 64), ("bf16", 2, 128)])
def test_composed_stem_block_second_order_equals_the_reference_double_backward(mode, N, S):
    """ops.DStemBwdFn: the MA-GP pattern on the composed stem block -- d(out . r)/d(image) with create_graph, a penalty on that
    gradient, and ITS gradients with respect to every parameter (train_gan.py:231-252) -- against autograd's double backward of
    conv_img + the block in f32 on the CPU, and against the un-composed second-order form (conv_img + ops.ResDFn) on the same inputs.
    The penalty is a plain sum of squares here (the sixth power of the reference amplifies rounding sixfold and is tested end to end)."""
    ops.set_precision(mode)
    dt = ops.act_dtype()
    from xmc_gan.model.df_gan import resD
    from xmc_gan.model.modules import HipConv2d
    torch.manual_seed(7)
    conv_img = HipConv2d(3, 32, 3, 1, 1).to(DEV)
    blk = resD(32, 64, True).to(DEV)
    with torch.no_grad():
        blk.gamma.fill_(0.6)
        conv_img.bias.normal_(0, 0.1)
        blk.conv_s.bias.normal_(0, 0.1)
    g = torch.Generator().manual_seed(N + S)
    x = rt(torch.rand(N, 3, S, S, generator=g) * 2 - 1, mode)
    r = rt(torch.randn(N, 64, S // 2, S // 2, generator=g), mode)
    params = [conv_img.weight, conv_img.bias, blk.conv_r[0].weight, blk.conv_r[2].weight, blk.conv_s.weight, blk.conv_s.bias, blk.gamma]
    r0, r2, s_ = blk.conv_r[0], blk.conv_r[2], blk.conv_s

    def penalty_grads(form):
        for p_ in params:
            p_.grad = None
        ops.new_iteration(DEV)
        xin = to_nhwc(x, 8, dt).requires_grad_()
        with ops.second_order():
            if form == "composed":
                out, _ = ops.DStemBlockFn.apply(xin, conv_img.weight, conv_img.bias, r0.weight, r2.weight, s_.weight, s_.bias, blk.gamma,
                                                conv_img.geom, r0.geom, r2.geom, s_.geom, True)
            else:
                ci, cip = conv_img(xin, want_pool=True)
                out, _ = blk(ci, xp_hint=cip, want_pool=True)
            with ops.no_wgrad():
                gx, = torch.autograd.grad((out.float() * to_nhwc(r, 64, torch.float32)).sum(), xin, create_graph=True)
        pen = (gx.float()[..., :3] ** 2).sum()
        pen.backward()
        return [None if p_.grad is None else p_.grad.detach().float().cpu().clone() for p_ in params], from_nhwc(gx.detach(), 3), pen.item()

    g_n, gx_n, pen_n = penalty_grads("composed")
    assert L.load().xmc_last_kernel().decode() != "", "no kernel ran"
    g_o, gx_o, pen_o = penalty_grads("uncomposed")
    P = [p_.detach().float().cpu().clone().requires_grad_() for p_ in params]
    xr = x.clone().requires_grad_()
    ci_r = F.conv2d(xr, P[0], P[1], 1, 1)
    br = F.leaky_relu(F.conv2d(F.leaky_relu(F.conv2d(ci_r, P[2], None, 2, 1), 0.2), P[3], None, 1, 1), 0.2)
    out_r = F.conv2d(F.avg_pool2d(ci_r, 2), P[4], P[5]) + P[6] * br
    gx_r, = torch.autograd.grad((out_r * r).sum(), xr, create_graph=True)
    pen_r = (gx_r ** 2).sum()
    pen_r.backward()
    tl = 3e-2 if mode == "bf16" else 4e-3
    ex_n, ex_o = rel_l2(gx_n, gx_r.detach()), rel_l2(gx_o, gx_r.detach())
    print(f"image gradient: composed vs f32 {ex_n:.2e}   conv_img + ResDFn vs f32 {ex_o:.2e};  penalty {pen_n:.5g} / {pen_o:.5g} / f32 {pen_r.item():.5g}")
    assert (ex_n < tl or ex_n < 1.5 * ex_o + 1e-3) and abs(pen_n - pen_r.item()) < 2 * tl * pen_r.item(), (ex_n, ex_o, pen_n, pen_r.item())
    names = ["conv_img.weight", "conv_img.bias", "conv_r.0.weight", "conv_r.2.weight", "conv_s.weight", "conv_s.bias", "gamma"]
    for n_, a, b, c in zip(names, g_n, g_o, P):
        if n_.endswith(".bias"):          # dx carries no bias term: zero (not None) gradients, as the reference's autograd hands out
            assert a is not None and float(a.abs().max()) == 0.0 and (c.grad is None or float(c.grad.abs().max()) == 0.0), n_
            continue
        e_ref, e_old = rel_l2(a, c.grad), rel_l2(b, c.grad)
        print(f"{n_:18s} composed vs f32 {e_ref:.2e}   conv_img + ResDFn vs f32 {e_old:.2e}")
        assert e_ref < 2 * tl or e_ref < 1.5 * e_old + 1e-3, (n_, e_ref, e_old)
    ops.set_precision("bf16")


def rel_l2(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def test_stem_composition_launches_equal_the_torch_statement():
    """xmc_dstem_compose / xmc_dstem_compose_bwd against ops.compose_dstem (the same algebra in differentiable torch ops) and its
    autograd adjoint: the four tables and the gradients of the five parameters for random table gradients."""
    g = torch.Generator().manual_seed(11)
    P = [torch.randn(32, 3, 3, 3, generator=g), torch.randn(32, generator=g), torch.randn(64, 32, 4, 4, generator=g) * 0.2,
         torch.randn(64, 32, 1, 1, generator=g), torch.randn(64, generator=g)]
    leaves = [p_.to(DEV).requires_grad_() for p_ in P]
    ref = ops.compose_dstem(*leaves)
    got = ops._dstem_compose_raw(*leaves)
    for a, b in zip(got, ref):
        assert a.shape == b.shape and rel_l2(a, b.detach()) < 1e-5
    douts = [torch.randn(t.shape, generator=g).to(DEV) for t in ref]
    for t in (douts[0], douts[2]):
        t[..., 3:] = 0                                  # the kernels never write gradient into the padded channels
    want = torch.autograd.grad(ref, leaves, douts)
    have = ops._dstem_compose_bwd_raw(*leaves, *douts)
    for name, a, b in zip(("conv_img.weight", "conv_img.bias", "conv_r.0.weight", "conv_s.weight", "conv_s.bias"), have, want):
        assert rel_l2(a.reshape(b.shape), b) < 1e-5, (name, rel_l2(a.reshape(b.shape), b))


@pytest.mark.parametrize("norm", [True, False])
def test_hoisted_sentence_queries_equal_the_per_stage_ones(norm):
    """ops.concept_query_all (every CondConceptSampler stage's sentence query in one launch, df_concept_gan.py:273-286) against the
    per-stage operator: values bit-equal; gradients of the sentence vector (the SUM over the stages), of every stage's projection and of
    its GroupNorm parameters to f32 summation order.  One stage is left without a consumer (no gradient reaches it)."""
    g = torch.Generator().manual_seed(7)
    B, E, S = 5, 256, 7
    sent = torch.randn(B, E, generator=g).to(DEV)
    stages = [(torch.nn.Parameter((torch.randn(64, E, 1, 1, generator=g) * 0.1).to(DEV)),
               torch.nn.Parameter((1 + 0.1 * torch.randn(64, generator=g)).to(DEV)) if norm else None,
               torch.nn.Parameter((0.1 * torch.randn(64, generator=g)).to(DEV)) if norm else None) for _ in range(S)]
    R = [torch.randn(B, 16, 4, generator=g).to(DEV) for _ in range(S)]
    s1 = sent.clone().requires_grad_()
    qs = ops.concept_query_all(s1, stages)
    sum((q * r).sum() for q, r in list(zip(qs, R))[:-1]).backward()          # the last stage's query is not used
    got = [(s1.grad.clone(),)] + [tuple(None if p is None or p.grad is None else p.grad.clone() for p in st) for st in stages]
    for st in stages:
        for p in st:
            if p is not None:
                p.grad = None
    s2 = sent.clone().requires_grad_()
    ref_q = [ops.concept_query(s2, *st) for st in stages]
    sum((q * r).sum() for q, r in list(zip(ref_q, R))[:-1]).backward()
    for a, b in zip(qs, ref_q):
        assert torch.equal(a, b)
    rel = lambda u, v: ((u - v).norm() / v.norm().clamp_min(1e-30)).item()
    assert rel(got[0][0], s2.grad) < 1e-5
    for s_, st in enumerate(stages):
        for k, p in enumerate(st):
            if p is None:
                continue
            if s_ == S - 1:
                assert p.grad is None and (got[1 + s_][k] is None or float(got[1 + s_][k].abs().max()) == 0.0)
            else:
                assert rel(got[1 + s_][k], p.grad) < 1e-5, (s_, k)


def test_integration_doc_snippet_runs_as_written():
    """The ctypes example of INTEGRATION.md (load the library, mirror XmcConvDesc, pack a weight, run a 3x3 convolution through the C ABI
    with nothing of this package imported) executed verbatim against F.conv2d."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(import ctypes as C, torch\n.*?)```", txt, re.S)
    assert m, "the ctypes example is gone from INTEGRATION.md"
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)                                  # the example opens the library by its in-tree relative path
    try:
        exec(m.group(1), ns)
    finally:
        os.chdir(cwd)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 16, 32, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(24, 32, 3, 3, generator=g) * 0.1).to(DEV)
    b = torch.randn(24, generator=g).to(DEV)
    y = ns["conv3x3"](x, w, b)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), b, padding=1).permute(0, 2, 3, 1)
    assert y.shape == (2, 16, 16, 24) and y.dtype == torch.bfloat16
    assert ((y.float() - ref).norm() / ref.norm()).item() < 5e-3


# ------------------------------------------------------------------------------------------ per-concept algebra (csrc/concept_word.hip, row a16)
def _close64(got, ref, tol=2e-5, what=""):
    got, ref = got.detach().double().cpu(), ref.detach().double()
    assert torch.equal(torch.isnan(got), torch.isnan(ref)), what
    m = ~torch.isnan(ref)
    err = (got[m] - ref[m]).norm() / ref[m].norm().clamp_min(1e-30)
    assert err <= tol, (what, err.item())
    return err.item()


@pytest.mark.parametrize("B,Is,Ig,O,bias", [(5, 356, 4, 8, True), (64, 0, 8, 4, False), (3, 0, 8, 4, True), (16, 356, 4, 8, False)])
def test_grouped_vector_convolution(B, Is, Ig, O, bias):
    """xmc_gvec_fwd / _bwd vs the reference's nn.Conv2d(groups=16) on cat(global condition, context) (concept_gan.py:346-371,404-418) in f64"""
    G = 16
    g = torch.Generator().manual_seed(B + Is)
    xs = torch.randn(B, Is, generator=g, dtype=torch.float64) if Is else None
    xg = torch.randn(B, G, Ig, generator=g, dtype=torch.float64)
    W = torch.randn(G * O, Is + Ig, 1, 1, generator=g, dtype=torch.float64) * 0.1
    bs = torch.randn(G * O, generator=g, dtype=torch.float64) if bias else None
    leaves = [t.clone().requires_grad_() for t in (xs, xg, W, bs) if t is not None]
    it = iter(leaves)
    xs_r = next(it) if xs is not None else None
    xg_r, W_r = next(it), next(it)
    bs_r = next(it) if bias else None
    cond = xg_r if xs_r is None else torch.cat([xs_r.view(B, 1, Is).expand(B, G, Is), xg_r], 2)
    ref = F.conv2d(cond.reshape(B, G * (Is + Ig), 1, 1), W_r, bs_r, groups=G).view(B, G, O)
    dy = torch.randn(B, G, O, generator=g, dtype=torch.float64)
    ref.backward(dy)
    dev = [None if t is None else t.float().to(DEV).requires_grad_() for t in (xs, xg, W, bs)]
    out = ops.grouped_vec(dev[0], dev[1], dev[2], dev[3], G)
    _close64(out, ref, what="forward")
    out.backward(dy.float().to(DEV))
    for name, d, r in (("xs", dev[0], xs_r), ("xg", dev[1], xg_r), ("W", dev[2], W_r), ("bias", dev[3], bs_r)):
        if d is not None:
            _close64(d.grad, r.grad, what=name)


@pytest.mark.parametrize("B", [1, 8, 300])
@pytest.mark.parametrize("bn_mode", ["none", "train", "eval"])
def test_concept_reasoner_with_batchnorm1d(B, bn_mode):
    """xmc_reasoner_fwd / _bwd vs concept_gan.ConceptReasoner (632-654) in f64: tanh adjacency, x + adj x, nn.BatchNorm1d(16) in training
    (batch statistics + running-statistics update) and in evaluation mode, relu; the whole batch in one launch."""
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 16, 4, generator=g, dtype=torch.float64)
    We = torch.randn(16, 4, generator=g, dtype=torch.float64) * 0.5
    bn64 = torch.nn.BatchNorm1d(16).double()
    with torch.no_grad():
        bn64.weight.copy_(torch.rand(16, generator=g, dtype=torch.float64) + 0.5)
        bn64.bias.copy_(torch.randn(16, generator=g, dtype=torch.float64) * 0.3)
        bn64.running_mean.copy_(torch.randn(16, generator=g, dtype=torch.float64) * 0.1)
        bn64.running_var.copy_(torch.rand(16, generator=g, dtype=torch.float64) + 0.5)
    bn64.train(bn_mode == "train")
    bn32 = torch.nn.BatchNorm1d(16).to(DEV)
    bn32.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in bn64.state_dict().items()})
    bn32.train(bn_mode == "train")
    if B == 1 and bn_mode == "train":
        pytest.skip("nn.BatchNorm1d refuses a single value per channel only when B * L == 1; B = 1 has L = 4 values, but the unbiased "
                    "variance of the running update is what differs -- covered by B = 8")
    xr, Wr = x.clone().requires_grad_(), We.clone().requires_grad_()
    adj = torch.tanh(F.linear(xr, Wr))
    pre = xr + torch.matmul(adj, xr)
    ref = F.relu(bn64(pre) if bn_mode != "none" else pre)
    dy = torch.randn(B, 16, 4, generator=g, dtype=torch.float64)
    ref.backward(dy)
    xd, Wd = x.float().to(DEV).requires_grad_(), We.float().to(DEV).requires_grad_()
    out = ops.reasoner(xd, Wd, bn32 if bn_mode != "none" else None)
    _close64(out, ref, what="forward")
    out.backward(dy.float().to(DEV))
    _close64(xd.grad, xr.grad, 5e-5, "dx")
    _close64(Wd.grad, Wr.grad, 5e-5, "dWe")
    if bn_mode != "none":
        _close64(bn32.weight.grad, bn64.weight.grad, 5e-5, "d bn.weight")
        _close64(bn32.bias.grad, bn64.bias.grad, 5e-5, "d bn.bias")
        _close64(bn32.running_mean, bn64.running_mean, what="running_mean")
        _close64(bn32.running_var, bn64.running_var, what="running_var")
        assert int(bn32.num_batches_tracked) == int(bn64.num_batches_tracked)


@pytest.mark.parametrize("B,T,case", [(6, 20, "ragged"), (4, 1, "single word"), (5, 15, "one caption all padding"), (3, 32, "no padding")])
def test_masked_word_attention_of_the_concepts(B, T, case):
    """xmc_word_ctx_fwd / _bwd vs OutConceptBlock.get_context_embs (concept_gan.py:374-394) in f64: states normalised over the CONCEPT
    axis, words over the state axis, masked_fill(-inf), softmax over T.  Padding words receive exactly zero gradient; a caption of
    padding only gives NaN in the same places as torch.softmax of an all -inf row; T = 1."""
    g = torch.Generator().manual_seed(T)
    st = torch.randn(B, 16, 4, generator=g, dtype=torch.float64)
    w = torch.randn(B, T, 4, generator=g, dtype=torch.float64)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    if case == "no padding":
        lens[:] = T
    mask = torch.arange(T)[None, :] >= lens[:, None]
    if case == "one caption all padding":
        mask[1, :] = True
    sr, wr = st.clone().requires_grad_(), w.clone().requires_grad_()
    sn, wd = F.normalize(sr, p=2, dim=1), F.normalize(wr, p=2, dim=2)
    sim = torch.matmul(sn, wd.transpose(1, 2)).masked_fill(mask.view(B, 1, -1), float("-inf"))
    ref = torch.matmul(torch.softmax(sim, dim=2), wd)
    dy = torch.randn(B, 16, 4, generator=g, dtype=torch.float64)
    ref.backward(dy)
    sd, wdv = st.float().to(DEV).requires_grad_(), w.float().to(DEV).requires_grad_()
    out = ops.word_context(sd, wdv, mask.to(DEV))
    _close64(out, ref, what="forward")
    out.backward(dy.float().to(DEV))
    ok = ~torch.isnan(ref).flatten(1).any(1)                                       # samples whose caption has at least one word
    _close64(sd.grad[ok.to(DEV)], sr.grad[ok], 5e-5, "d state")
    _close64(wdv.grad[ok.to(DEV)], wr.grad[ok], 5e-5, "d words")
    assert (wdv.grad.cpu()[ok][mask[ok]] == 0).all()                               # padding words: exactly zero
    if case == "one caption all padding":
        assert torch.isnan(out[1]).all() and not torch.isnan(out[0]).any()


@pytest.mark.parametrize("B,T,norm", [(5, 20, True), (2, 1, True), (7, 15, False)])
def test_word_keys_groupnorm_and_normalisation(B, T, norm):
    """xmc_word_keys_fwd / _bwd vs CondConceptSampler's key path (concept_gan.py:566-575) in f64: [B,64,T] -> GroupNorm(16, 64) over
    (state, word) -> L2 normalisation over the state axis -> [B,16,T,4]"""
    g = torch.Generator().manual_seed(B * T)
    kraw = torch.randn(B, T, 64, generator=g, dtype=torch.float64)
    gw, gb = torch.rand(64, generator=g, dtype=torch.float64) + 0.5, torch.randn(64, generator=g, dtype=torch.float64) * 0.2
    kr, gwr, gbr = kraw.clone().requires_grad_(), gw.clone().requires_grad_(), gb.clone().requires_grad_()
    k = kr.transpose(1, 2)
    if norm:
        k = F.group_norm(k, 16, gwr, gbr, 1e-5)
    ref = F.normalize(k.reshape(B, 16, 4, T), p=2, dim=2).permute(0, 1, 3, 2)
    dy = torch.randn(B, 16, T, 4, generator=g, dtype=torch.float64)
    ref.backward(dy)
    kd = kraw.float().to(DEV).requires_grad_()
    gwd, gbd = gw.float().to(DEV).requires_grad_(), gb.float().to(DEV).requires_grad_()
    out = ops.word_keys(kd, gwd if norm else None, gbd if norm else None, 1e-5)
    _close64(out, ref, what="forward")
    out.backward(dy.float().to(DEV))
    _close64(kd.grad, kr.grad, 1e-4, "d kraw")
    if norm:
        _close64(gwd.grad, gwr.grad, 5e-5, "d gn.weight")
        _close64(gbd.grad, gbr.grad, 5e-5, "d gn.bias")
