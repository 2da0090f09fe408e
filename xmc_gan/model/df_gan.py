"""DF-GAN generator / discriminator with the reference's module API (model/df_gan.py), running on the
MI355X kernels in ``xmc-gan_amd``.

Same class names, constructor signatures ``(cfg, **kwargs)``, forward signatures, return conventions
and ``state_dict()`` keys as the reference, so it is a drop-in for ``train_gan.py``'s registries.
Internally activations are NHWC in the engine's activation dtype; the public tensors keep the
reference's logical NCHW shapes (``netD(x)`` returns a channels-last view, ``netG`` an f32 NCHW image).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from xmc_gan_amd import ops
from xmc_gan_amd.lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH

from .modules import HipConv2d, HipLinear, as_nchw_view, as_nhwc, conv2d_nxn, linear


def gen_arch(img_size, nch):
    """channel/upsample schedule of the generator (reference table: df_gan.py:9-34)."""
    assert img_size in [64, 128, 256]
    depth = {64: 5, 128: 6, 256: 7}[img_size]
    mult = [8] * (depth - 2) + [4, 2, 1]
    return {
        'in_channels': [m * nch for m in mult[:-1]],
        'out_channels': [m * nch for m in mult[1:]],
        'upsample': [True] * (depth - 1) + [False],
        'resolution': [8 << i for i in range(depth - 1)] + [img_size],
        'depth': depth,
    }


def disc_arch(img_size, nch):
    """channel schedule of the discriminator (reference table: df_gan.py:36-61)."""
    assert img_size in [64, 128, 256]
    depth = {64: 5, 128: 6, 256: 7}[img_size]
    mult = [1, 2, 4, 8, 16, 16, 16][:depth]
    out_channels = [m * nch for m in mult]
    return {
        'in_channels': [3] + out_channels[:-1],
        'out_channels': out_channels,
        'downsample': [True] * depth,
        'resolution': [img_size >> (i + 1) for i in range(depth - 1)] + [4],
        'depth': depth,
    }


def nhwc_feature_perm(channels, hw=16):
    """row permutation that makes Linear(...)->view(B,C,4,4) come out as NHWC [B,4,4,C] directly:
    packed row r = p*C + c  <-  parameter row c*hw + p."""
    return [(r % channels) * hw + r // channels for r in range(channels * hw)]


class NetG(nn.Module):
    nhwc_out = True       # forward(..., return_nhwc=True) -> (image NCHW f32, the same image in the engine layout)

    def __init__(self, cfg, **kwargs):
        super(NetG, self).__init__()
        self.ngf = cfg.TRAIN.NCH
        noise_dim = cfg.TRAIN.NOISE_DIM
        arch = gen_arch(img_size=cfg.IMG.SIZE, nch=self.ngf)

        init_size = (8 * self.ngf) * 4 * 4
        self.proj_noise = HipLinear(noise_dim, init_size, row_perm=nhwc_feature_perm(8 * self.ngf))
        self.proj_sent = HipLinear(cfg.TEXT.EMBEDDING_DIM, cfg.TRAIN.NEF) \
            if (cfg.TEXT.EMBEDDING_DIM != cfg.TRAIN.NEF) else nn.Identity()
        self.upblocks = nn.ModuleList(
            [G_Block(in_dim=arch['in_channels'][i], out_dim=arch['out_channels'][i],
                     cond_dim=cfg.TRAIN.NEF, upsample=arch['upsample'][i]) for i in range(arch['depth'])])
        self.conv_out = nn.Sequential(
            nn.LeakyReLU(0.2, inplace=True),
            HipConv2d(arch['out_channels'][-1], 3, 3, 1, 1),
            nn.Tanh(),
        )

    def stem(self, noise):
        """proj_noise + view(B, 8*ngf, 4, 4) (reference df_gan.py:93-94), emitted as NHWC."""
        out = self.proj_noise(noise.float(), out_dtype=ops.act_dtype())
        return out.view(noise.size(0), 4, 4, 8 * self.ngf)

    nhwc_dst_ok = True       # forward(..., nhwc_dst=) writes the engine-layout image into a caller-provided tensor

    def tail(self, out, lrelu_done=False, return_nhwc=False, nhwc_dst=None):
        """LeakyReLU -> Conv3x3(->3) -> Tanh (reference df_gan.py:84-88,101); tanh fused in the conv epilogue.
        ``return_nhwc``: also hand out the engine-layout image [B,S,S,8] the NCHW f32 result was converted from, so a
        discriminator call on this image (``netD(fake, nhwc8=...)``) skips the NHWC->NCHW->NHWC round trip in both directions.
        ``nhwc_dst``: [B,S,S,8] tensor the engine-layout image is written into (one half of the discriminator's 2B input)."""
        if not lrelu_done:
            out = ops.lrelu(out)
        out = self.conv_out[1](out, act=ACT_TANH, out=nhwc_dst)
        img = ops.to_nchw(out, 3)
        return (img, out) if return_nhwc else img

    def forward(self, noise, sent_embs, return_nhwc=False, nhwc_dst=None, **kwargs):
        out = self.stem(noise)
        sent_embs = self.proj_sent(sent_embs.float())
        # The 8 conditioning MLPs of every block depend only on the sentence embedding: all of them (40-56 two-layer MLPs)
        # run as grouped GEMM launches up front.
        nblk = len(self.upblocks)
        flat = ops.cond_mlp_bank(sent_embs, [m for g in self.upblocks for m in g.modulation_mlps()])
        mods = [flat[8 * i:8 * i + 8] for i in range(nblk)]
        # Blocks hand over their output BEFORE the nearest x2 upsample (df_gan.py:201-202); the next block consumes it
        # through operators that commute with / absorb the upsample, so the 4x larger tensor is never written.
        pending_up = False
        lrelu_done = False
        from_end = False            # `out` was written by an ops.GBlockEndFn (whose backward wants the 2x2 sum pool of its gradient)
        for bi, (gblock, m) in enumerate(zip(self.upblocks, mods)):
            last = bi == nblk - 1 and not gblock.upsample      # its output goes straight into the tail's LeakyReLU
            fuse = gblock.fuses_end(out, pending_up)
            if last and fuse:
                # the block's second half and the tail (LeakyReLU -> conv_out -> tanh) as one node: ops.GBlockEndFn
                co = self.conv_out[1]
                out8 = gblock.forward_fused(out, m, pending_up, out_lrelu=True, tail=(co.weight, co.bias, co.geom), nhwc_dst=nhwc_dst,
                                            x_from_end=from_end)
                img = ops.to_nchw(out8, 3)
                return (img, out8) if return_nhwc else img
            out = gblock.forward_fused(out, m, pending_up, out_lrelu=last, x_from_end=from_end)
            from_end = fuse and not last
            lrelu_done = last
            pending_up = gblock.upsample
        if pending_up:
            out = ops.upsample2(out)
        return self.tail(out, lrelu_done, return_nhwc, nhwc_dst)


class NetD(nn.Module):
    def __init__(self, cfg, **kwargs):
        super(NetD, self).__init__()
        ndf = cfg.TRAIN.NCH
        spec_norm = cfg.DISC.SPEC_NORM
        arch = disc_arch(img_size=cfg.IMG.SIZE, nch=ndf)
        self.conv_img = conv2d_nxn(in_dim=arch['in_channels'][0], out_dim=arch['out_channels'][0], kernel_size=3,
                                   stride=1, padding=1, spec_norm=spec_norm)
        self.downblocks = nn.ModuleList(
            [resD(in_dim=arch['in_channels'][i], out_dim=arch['out_channels'][i],
                  downsample=arch['downsample'][i], spec_norm=spec_norm) for i in range(1, arch['depth'])])
        self.COND_DNET = D_GET_LOGITS(cfg, ndf=ndf, spec_norm=spec_norm)

    def forward(self, x, nhwc8=None, **kwargs):
        """x: [B,3,S,S] f32 image -> [B,16*ndf,4,4] feature map (channels-last view).  ``nhwc8``: the same image already in
        the engine layout [B,S,S,8] (``ops.to_nhwc8(x)`` or NetG's ``return_nhwc`` output); ``x`` is then not read."""
        pooled = None               # avg_pool2d of `out`, written by the layer that produced it (third output of its epilogue)
        xin = ops.to_nhwc8(x) if nhwc8 is None else nhwc8
        # A LEAF input that requires grad is the gradient-penalty pattern (train_gan.py:231-237: `imgs.detach().requires_grad_()`,
        # then autograd.grad(..., create_graph=True)): the blocks then keep what a differentiated backward needs, as inside
        # ops.second_order().  The training passes feed images without grad (D step) or the generator's output (G step, not a leaf).
        src = x if nhwc8 is None else nhwc8
        leaf_in = torch.is_tensor(src) and src.requires_grad and src.is_leaf
        with ops.second_order(leaf_in or ops.second_order_active()):
            nblk = len(self.downblocks)
            b0 = self.downblocks[0]
            first = 0
            sink, cut = self.cut_sink, self.cut_block()
            # conv_img + the first block on the composed stem (ops.DStemBlockFn: the image straight to the block's first activation and
            # to its shortcut; conv_img's output never exists).  Under MA-GP (the backward of this forward is differentiated again)
            # the node keeps the branch values and its backward is ops.DStemBwdFn, on the same stem kernels
            if (ops.fused_blocks() and not self.conv_img.spec_norm and b0.downsample
                    and not (ops.second_order_active() and ops.debug_switch("no_dstem2"))
                    and ops.dstem_eligible(xin, self.conv_img.out_channels, b0.learned_shortcut, b0.conv_r[0].out_channels)):
                r0, r2, s_ = b0.conv_r[0], b0.conv_r[2], b0.conv_s
                out, pooled = ops.DStemBlockFn.apply(xin, self.conv_img.weight, self.conv_img.bias, r0.weight, r2.weight, s_.weight, s_.bias,
                                                     b0.gamma, self.conv_img.geom, r0.geom, r2.geom, s_.geom, nblk > 1)
                first = 1
            elif ops.fused_blocks() and xin.is_cuda and xin.shape[1] % 2 == 0:
                out, pooled = self.conv_img(xin, want_pool=True)
            else:
                out = self.conv_img(xin)
            for i, block in enumerate(self.downblocks):
                if i < first:
                    if sink is not None and i == cut:         # the cut is the composed stem's output
                        sink.extend(t for t in (out, pooled) if torch.is_tensor(t) and t.requires_grad)
                    continue
                out, pooled = block(out, xp_hint=pooled, want_pool=i + 1 < nblk)
                if sink is not None and i == cut:
                    sink.extend(t for t in (out, pooled) if torch.is_tensor(t) and t.requires_grad)
        return as_nchw_view(out)

    # Data parallel (xmc_gan/train_gan.py: the discriminator step): the trunk handed out where it enters the last three blocks, so that the
    # backward can be run in two parts -- head and last blocks first (most of the parameter bytes, a fraction of the time), their
    # gradient all-reduce started, then the rest beside it.  `cut_sink` is a list while a caller wants those tensors, else None.
    cut_sink = None

    def cut_block(self):
        """index of the block whose output is the cut (None: too few blocks, or spectral norm -- the composed blocks are left alone)"""
        k = len(self.downblocks) - 4
        return k if k >= 0 and not self.conv_img.spec_norm else None

    def late_parameters(self):
        """(parameters after the cut, parameters before it), each in registration order"""
        k = self.cut_block()
        early = list(self.conv_img.parameters()) + [p for b in self.downblocks[:k + 1] for p in b.parameters()]
        ids = {id(p) for p in early}
        return [p for p in self.parameters() if id(p) not in ids], early


class D_GET_LOGITS(nn.Module):
    """Conditional logit and contrastive projection head (reference df_gan.py:134-176)."""

    def __init__(self, cfg, ndf, spec_norm=False):
        super(D_GET_LOGITS, self).__init__()
        nef = cfg.TRAIN.NEF
        text_dim = cfg.TEXT.EMBEDDING_DIM
        self.img_match = cfg.DISC.IMG_MATCH
        if self.img_match:
            self.proj_match = linear(ndf * 16, nef, spec_norm=spec_norm)       # image side
            cond_dim = nef
        elif cfg.DISC.SENT_MATCH:
            self.proj_match = linear(nef, ndf * 16, spec_norm=spec_norm)       # sentence side
            cond_dim = ndf * 16
        elif cfg.DISC.SEPERATE and (text_dim != nef):
            self.proj_match = linear(text_dim, nef, spec_norm=spec_norm)
            cond_dim = nef
        else:
            self.proj_match = nn.Identity()
            cond_dim = text_dim
        self.joint_conv = nn.Sequential(
            conv2d_nxn(in_dim=ndf * 16 + cond_dim, out_dim=ndf * 2, kernel_size=3, stride=1, padding=1, bias=False,
                       spec_norm=spec_norm),
            nn.LeakyReLU(0.2, inplace=True),
            conv2d_nxn(in_dim=ndf * 2, out_dim=1, kernel_size=4, stride=1, padding=0, bias=False, spec_norm=spec_norm),
        )

    def forward(self, x, sent_embs, **kwargs):
        """x [B,16*ndf,4,4], sent_embs [B,cond] -> [logit [B,1,1,1], image embedding, text embedding]."""
        xh = as_nhwc(x)
        B = xh.size(0)
        out = ops.global_avgpool(xh)                               # F.avg_pool2d(x, 4).view(B, -1)
        sent_embs = sent_embs.float()
        if self.img_match:
            out = self.proj_match(out)
        else:
            sent_embs = self.proj_match(sent_embs)
        c = ops.cast(_pad8(sent_embs), xh.dtype)
        c = c.view(B, 1, 1, -1).expand(B, xh.size(1), xh.size(2), c.size(-1))
        h_c_code = torch.cat((xh, c), 3)
        j0 = self.joint_conv[0]
        if h_c_code.dtype == torch.float32 and ops.act_dtype() != torch.float32 and not j0.spec_norm and j0.bias is None:
            # precise trunk (the last block handed over an f32 map): f32 grade on the 16-bit matrix pipeline (ops.PairConvFn)
            m = ops.pair_conv2d(h_c_code, j0.weight, j0.geom, ACT_LRELU)
        else:
            m = j0(h_c_code, act=ACT_LRELU)
        m = self.joint_conv[2](m, out_dtype=torch.float32)         # [B,1,1,8] f32, channel 0 is the logit (losses are f32)
        match = as_nchw_view(m[..., :1])
        return [match, out, sent_embs]


def _pad8(t):
    r = (-t.size(-1)) % 8
    return t if r == 0 else torch.nn.functional.pad(t, (0, r))


class G_Block(nn.Module):
    def __init__(self, in_dim, out_dim, cond_dim, upsample):
        super(G_Block, self).__init__()
        self.learnable_sc = (in_dim != out_dim)
        self.upsample = upsample
        self.c1 = HipConv2d(in_dim, out_dim, 3, 1, 1)
        self.c2 = HipConv2d(out_dim, out_dim, 3, 1, 1)
        self.affine0 = affine(num_features=in_dim, cond_dim=cond_dim)
        self.affine1 = affine(num_features=in_dim, cond_dim=cond_dim)
        self.affine2 = affine(num_features=out_dim, cond_dim=cond_dim)
        self.affine3 = affine(num_features=out_dim, cond_dim=cond_dim)
        self.gamma = nn.Parameter(torch.zeros(1))
        if self.learnable_sc:
            self.c_sc = HipConv2d(in_dim, out_dim, 1, stride=1, padding=0)

    def modulation_mlps(self):
        """(w1, b1, w2, b2) of the eight conditioning MLPs, in the order ``modulation`` returns their outputs"""
        out = []
        for a in (self.affine0, self.affine1, self.affine2, self.affine3):
            for m in (a.fc_gamma, a.fc_beta):
                out.append((m.linear1.weight, m.linear1.bias, m.linear2.weight, m.linear2.bias))
        return out

    def modulation(self, c):
        """the eight per-sample (scale, shift) vectors of this block, f32 [B,C] each"""
        return (*self.affine0.scale_shift(c), *self.affine1.scale_shift(c),
                *self.affine2.scale_shift(c), *self.affine3.scale_shift(c))

    def forward(self, x, c, mod=None):
        out = ops.axpby(self.shortcut(x), self.residual(x, c, mod), self.gamma)
        if self.upsample:
            out = ops.upsample2(out)
        return out

    def fuses_end(self, x, x_pending_up):
        """whether forward_fused runs the block's second half (and, for the last block, the network's tail) as ops.GBlockEndFn"""
        return bool(x_pending_up) and ops.fused_blocks() and x.is_cuda

    def forward_fused(self, x, mod, x_pending_up, out_lrelu=False, tail=None, nhwc_dst=None, x_from_end=False):
        """Same function as ``forward`` on the logical input ``up2(x)`` when ``x_pending_up`` (else ``x``), returning the
        block output WITHOUT its trailing upsample.  With a pending upsample: the conditional affines and the 1x1 shortcut
        commute with nearest upsampling and run at low resolution, ``c1`` runs as the fused upsample+3x3 operator
        (4 parity classes of 2x2 taps: 4/9 of the MACs) and the shortcut is upsampled inside the final add."""
        if not x_pending_up:
            out = ops.axpby(self.shortcut(x), self.residual(x, None, mod), self.gamma)
            return ops.lrelu(out) if out_lrelu else out
        # x feeds the residual branch AND the shortcut: the affine node hands x through as a second output, so that the two
        # gradients of x are summed inside its backward kernel
        h, xs = ops.affine2_lrelu_skip(x, *mod[0:4], pool_grad=x_from_end) if x.is_cuda else (ops.affine2_lrelu(x, *mod[0:4]), x)
        h = ops.upconv3x3(h, self.c1.weight, self.c1.bias, self.c1.geom)
        if self.fuses_end(x, x_pending_up) and (tail is not None or not out_lrelu):
            # affine2/3 + LeakyReLUs, c2, the block sum with the upsampled shortcut [, the tail] in one node whose backward
            # needs neither c2's output nor a separate pass for gamma * dout (ops.GBlockEndFn)
            return ops.g_block_end(h, mod[4:8], self.c2.weight, self.c2.bias, self.c2.geom, self.shortcut(xs), self.gamma,
                                   tail=tail, nhwc_dst=nhwc_dst)
        assert tail is None
        h = ops.affine2_lrelu(h, *mod[4:8])
        if not out_lrelu and ops.fused_blocks() and h.shape[1] % 2 == 0:
            # c2, the block sum and the upsample of the shortcut in one pass (third epilogue form, res_mode 2)
            return ops.conv_axpby_up(h, self.c2.weight, self.c2.bias, self.c2.geom, self.shortcut(xs), self.gamma)
        # out_lrelu: the tail's LeakyReLU (df_gan.py:84-85) applied while the block sum is written
        return ops.axpby_up(self.shortcut(xs), self.c2(h), self.gamma, lrelu=out_lrelu)

    def shortcut(self, x):
        # precise trunk (IEEE-half mode, ops.precise_trunk): the image is, to first order in the block gammas, a function of the shortcut
        # path alone, and a rounding error of c_sc's weights is shared by every sample and pixel -- forward on the weights' hi + lo pair
        return self.c_sc(x, pair=ops.precise_trunk()) if self.learnable_sc else x

    def residual(self, x, c, mod=None):
        m = self.modulation(c) if mod is None else mod
        # affine0 -> LeakyReLU -> affine1 -> LeakyReLU fused into one pass (df_gan.py:213-216), same for 2/3
        h = ops.affine2_lrelu(x, *m[0:4])
        h = self.c1(h)
        h = ops.affine2_lrelu(h, *m[4:8])
        return self.c2(h)


class _CondMLP(nn.Module):
    """Linear(cond,256) -> ReLU -> Linear(256,C) with the reference's child names (df_gan.py:232-241)."""

    def __init__(self, cond_dim, num_features):
        super().__init__()
        self.linear1 = HipLinear(cond_dim, 256)
        self.relu1 = nn.ReLU(inplace=True)
        self.linear2 = HipLinear(256, num_features)

    def forward(self, y):
        return self.linear2(self.linear1(y, act=ACT_RELU))


class affine(nn.Module):
    def __init__(self, num_features, cond_dim):
        super(affine, self).__init__()
        self.fc_gamma = _CondMLP(cond_dim, num_features)
        self.fc_beta = _CondMLP(cond_dim, num_features)
        self._initialize()

    def _initialize(self):
        nn.init.zeros_(self.fc_gamma.linear2.weight.data)
        nn.init.ones_(self.fc_gamma.linear2.bias.data)
        nn.init.zeros_(self.fc_beta.linear2.weight.data)
        nn.init.zeros_(self.fc_beta.linear2.bias.data)

    def scale_shift(self, y):
        """per-sample, per-channel (weight, bias), each f32 [B,C]."""
        return self.fc_gamma(y), self.fc_beta(y)

    def forward(self, x, y=None):
        """weight(y) * x + bias(y) on an NHWC tensor (reference df_gan.py:250-263)."""
        w, b = self.scale_shift(y)
        return x * w[:, None, None, :].to(x.dtype) + b[:, None, None, :].to(x.dtype)


class resD(nn.Module):
    def __init__(self, in_dim, out_dim, downsample, spec_norm=False):
        super().__init__()
        self.downsample = downsample
        self.learned_shortcut = (in_dim != out_dim)
        self.conv_r = nn.Sequential(
            conv2d_nxn(in_dim=in_dim, out_dim=out_dim, kernel_size=4, stride=2, padding=1, bias=False, spec_norm=spec_norm),
            nn.LeakyReLU(0.2, inplace=True),
            conv2d_nxn(in_dim=out_dim, out_dim=out_dim, kernel_size=3, stride=1, padding=1, bias=False, spec_norm=spec_norm),
            nn.LeakyReLU(0.2, inplace=True),
        )
        self.conv_s = conv2d_nxn(in_dim=in_dim, out_dim=out_dim, kernel_size=1, stride=1, padding=0, spec_norm=spec_norm)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x, c=None, xp_hint=None, want_pool=None):
        """`want_pool` given (NetD's loop): returns (out, avg_pool2d(out, 2) or None); `xp_hint`: avg_pool2d(x, 2) if the previous
        block already produced it.  Called the upstream way (`block(x)`), returns `out` alone."""
        if self.downsample and ops.fused_blocks() and x.is_cuda:
            r0, r2, s_ = self.conv_r[0], self.conv_r[2], self.conv_s
            r = ops.ResDFn.apply(x, r0.effective_weight(), r2.effective_weight(),
                                 s_.effective_weight() if self.learned_shortcut else None,
                                 s_.bias if self.learned_shortcut else None, self.gamma, r0.geom, r2.geom, s_.geom,
                                 xp_hint, bool(want_pool))
            if want_pool is None:
                return r
            return r if want_pool else (r, None)
        out = ops.axpby(self.shortcut(x), self.residual(x), self.gamma)
        return out if want_pool is None else (out, None)

    def shortcut(self, x):
        # upstream: avg_pool2d(conv_s(x), 2) (df_gan.py:286-291).  A 1x1 convolution (and its bias) commutes with
        # average pooling, so pooling first gives the same function with 4x less conv work and HBM traffic.
        if self.downsample:
            x = ops.avgpool2(x)
        if self.learned_shortcut:
            x = self.conv_s(x)
        return x

    def residual(self, x):
        r = self.conv_r[0](x, act=ACT_LRELU)        # conv + LeakyReLU fused in the epilogue
        return self.conv_r[2](r, act=ACT_LRELU)
