"""Word-attention generator with the reference's module API (model/concept_gan.py): ``OutNetG`` -- two BatchNorm-conditional
``ResBlockUp`` stages followed by ``OCAttnResBlockUp`` stages whose ``OutConceptBlock`` lets 16 image concepts attend over
the (masked) caption words.  Upstream leaves these classes out of ``_GEN_ARCH`` (their names are commented out,
train_gan.py:31,44); here ``OutNetG`` is registered as ``CONCEPT_OUTATTN_GEN`` and ``InNetG`` as ``CONCEPT_INATTN_GEN``.

``InNetG`` -- the same stem and ``ResBlockUp`` stages followed by ``ICAttnResBlockUp`` stages in which every REGION of the map attends
over the caption words -- cannot run upstream; it is built here as a documented REPAIR of exactly the two defects that stop it
(SURVEY 2c): (1) ``CondConceptSampler.key_gconv`` is constructed for ``noise_dim + nef`` input channels per concept (concept_gan.py:
137,183,527) but is fed the projected words, ``nef`` channels (570-573): it takes ``nef`` here, which is the only ``state_dict``
shape that differs from what upstream's constructor registers; (2) ``InConceptBlock.forward`` reads ``self.upsample`` (222), which
no constructor assigns: the block inherits the flag of the enclosing ``ICAttnResBlockUp``, the value that block's shortcut (148-149)
needs the residual to agree with.  Nothing else is changed; ``oracle/make_golden.py`` applies the same two patches to the reference's
own objects and the fixtures ``fwd_wordin*`` / ``step_wordin*`` pin the result.

Same constructor/forward signatures and ``state_dict()`` keys (incl. the ``linaer_beta2`` spelling and the BatchNorm
buffers).  Per-pixel work runs on the HIP kernels (MFMA convolutions incl. the fused upsample+3x3, BatchNorm/GroupNorm,
region attention, conditional modulation) and so does the per-sample concept algebra on [B,16,<=360] tensors (since round 5:
reasoner with its BatchNorm1d, masked word attention over T <= 32 words, grouped 1x1 modulation heads, the word keys'
GroupNorm -- csrc/concept_word.hip); what is left to ATen is batch-sized glue (the cat of noise and sentence, reshapes).

Behaviour kept because it changes results (concept_gan.py): the first two blocks receive the whole ``upsample`` list as
their flag, i.e. both upsample (262); in ``OutConceptBlock`` the second sampler's output is discarded and the second reasoner
is applied to the first context only to be discarded as well (431-433) -- all that survives is its BatchNorm1d
running-statistics update, which is reproduced; the state vectors are L2-normalised over the CONCEPT axis (378).
Restructured, same function: everything after the block's grouped 3x3 conv is pointwise or 1x1 and therefore commutes
with the nearest x2 upsample, so the block runs at the input resolution and is upsampled once at the end.  In ``InConceptBlock`` the
second sampler reads the upsampled map (222-227): its GroupNorm statistics, its per-region attention and the mean over regions are
all unchanged by replicating every region four times, so that stage runs at the input resolution too.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from xmc_gan_amd import ops
from xmc_gan_amd.lib import ACT_RELU, ACT_TANH

from .df_concept_gan import ConceptSampler, _GroupedConv
from .df_gan import nhwc_feature_perm
from .modules import HipConv2d, HipLinear


def gen_arch(img_size, nch):
    """channel / resolution tables (concept_gan.py:11-37)"""
    assert img_size in [64, 128, 256]
    mult_in, mult_out = {256: ([16, 16, 8, 8, 4, 2, 1], [16, 8, 8, 4, 2, 1, 1]),
                         128: ([16, 8, 8, 4, 2, 1], [8, 8, 4, 2, 1, 1]),
                         64: ([8, 8, 4, 2, 1], [8, 4, 2, 1, 1])}[img_size]
    depth = len(mult_in)
    res = [8 * 2 ** i for i in range(depth - 1)]
    return {'in_channels': [m * nch for m in mult_in], 'out_channels': [m * nch for m in mult_out],
            'upsample': [True] * (depth - 1) + [False], 'resolution': res + [res[-1]],
            'attention': [False] * 2 + [True] * (depth - 2), 'depth': depth}


class _Pad8Linear(HipLinear):
    """nn.Linear parameters whose input width is not a multiple of 8 (noise_dim + nef = 356): the kernels' data-gradient
    path wants 8-channel units, so input and weight are zero-padded to the next multiple of 8 on the way in (the gradient
    flows back through the pad to the [out, in] parameter)."""

    def __init__(self, in_dim, out_dim, bias=True, row_perm=None):
        super().__init__(in_dim, out_dim, bias=bias, row_perm=row_perm)
        self._pad = (-in_dim) % 8
        self.geom = ops.ConvGeom(in_dim + self._pad, out_dim, 1, 1, 0, row_perm=row_perm)

    def forward(self, x, act=None, out_dtype=None):
        w = self.weight
        if self._pad:
            x, w = F.pad(x, (0, self._pad)), F.pad(w, (0, self._pad))
        y = ops.linear(x, w, self.bias, self.geom, self.act if act is None else act, out_dtype)
        return y[:, : self.out_features] if y.shape[1] != self.out_features else y


_ONES = {}


def _one(device):
    t = _ONES.get(device)
    if t is None:
        t = _ONES[device] = torch.ones(1, dtype=torch.float32, device=device)
    return t


def _cond_bn_relu(x, bn, gamma, beta):
    """relu(gamma[n,c] * BatchNorm(x) + beta[n,c]) on NHWC (concept_gan.py:494-497, 506-509); bn None = no normalisation."""
    if bn is None:
        return ops.affine_act(x, gamma, beta, 0.0)
    if bn.training:
        y, stats = ops.batchnorm_train(x, bn.weight, bn.bias, bn.eps)
        with torch.no_grad():       # running statistics as nn.BatchNorm2d keeps them: momentum 0.1, unbiased variance
            n = x.numel() // x.shape[-1]
            mean, var = stats[:, 0], stats[:, 1].pow(-2) - bn.eps
            bn.running_mean.mul_(1 - bn.momentum).add_(mean, alpha=bn.momentum)
            bn.running_var.mul_(1 - bn.momentum).add_(var * (n / max(n - 1, 1)), alpha=bn.momentum)
            bn.num_batches_tracked += 1
        return ops.affine_act(y, gamma, beta, 0.0)
    # eval: BatchNorm is a per-channel scale/shift, folded into the conditional one
    s = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
    t = bn.bias.float() - bn.running_mean.float() * s
    return ops.affine_act(x, gamma * s, gamma * t + beta, 0.0)


class ResBlockUp(nn.Module):
    def __init__(self, in_dim, out_dim, cond_dim, upsample, normalize=True):
        super(ResBlockUp, self).__init__()
        self.learnable_sc = (in_dim != out_dim)
        self.normalize = normalize
        self.upsample = upsample
        self.c1 = HipConv2d(in_dim, out_dim, 3, 1, 1)
        self.c2 = HipConv2d(out_dim, out_dim, 3, 1, 1)
        if normalize:
            self.bn1 = nn.BatchNorm2d(in_dim)
            self.bn2 = nn.BatchNorm2d(out_dim)
        self.linear_gamma1 = _Pad8Linear(cond_dim, in_dim, bias=False)
        self.linear_beta1 = _Pad8Linear(cond_dim, in_dim, bias=False)
        self.linear_gamma2 = _Pad8Linear(cond_dim, out_dim, bias=False)
        self.linaer_beta2 = _Pad8Linear(cond_dim, out_dim, bias=False)          # sic (concept_gan.py:473)
        if self.learnable_sc:
            self.c_sc = HipConv2d(in_dim, out_dim, 1, stride=1, padding=0)

    def forward(self, x, global_cond, **kwargs):
        """x NHWC [B,h,w,Cin] -> [B,2h,2w,Cout] (or same size when ``upsample`` is falsy)."""
        h = _cond_bn_relu(x, self.bn1 if self.normalize else None,
                          self.linear_gamma1(global_cond), self.linear_beta1(global_cond))
        if self.upsample:       # F.interpolate(x2) -> c1 as one operator (4 parity classes of 2x2 taps)
            h = ops.upconv3x3(h, self.c1.weight, self.c1.bias, self.c1.geom)
        else:
            h = self.c1(h)
        h = _cond_bn_relu(h, self.bn2 if self.normalize else None,
                          self.linear_gamma2(global_cond), self.linaer_beta2(global_cond))
        r = self.c2(h)
        sc = self.c_sc(x) if self.learnable_sc else x                         # the 1x1 shortcut commutes with the upsample
        one = _one(x.device)
        return ops.axpby_up(sc, r, one) if self.upsample else ops.axpby(sc, r, one)


class ConceptReasoner(nn.Module):
    """concept graph step on [B,16,p'] states (concept_gan.py:632-654); BatchNorm1d over the concepts is live here."""

    def __init__(self, cardinality, state_dim, normalize=True):
        super(ConceptReasoner, self).__init__()
        self.cardinality = cardinality
        self.normalize = normalize
        self.proj_edge = nn.Linear(state_dim, cardinality, bias=False)
        if self.normalize:
            self.bn = nn.BatchNorm1d(num_features=cardinality)

    def forward(self, x, **kwargs):
        """x [B,16,4] f32 -> relu(BatchNorm1d(x + tanh(x We^T) x)): one launch for the whole batch (csrc/concept_word.hip), the running
        statistics updated in place in training mode like nn.BatchNorm1d (momentum 0.1, unbiased variance)"""
        return ops.reasoner(x, self.proj_edge.weight, self.bn if self.normalize else None)


def _word_context(state, words, mask):
    """state [B,C,p'], words [B,T,p'], mask [B,T] (True = padding) -> attention of every concept over the words
    (OutConceptBlock.get_context_embs, concept_gan.py:374-394): [B,C,p'].  States are L2-normalised over the CONCEPT axis, as upstream."""
    return ops.word_context(state, words, mask)


def _project_words(words_embs, conv1d):
    """nn.Conv1d(text_dim, n, 1) on words [B,T,text_dim] as one GEMM over the B * T words -> [B,T,n]"""
    B, T, E = words_embs.shape
    n = conv1d.out_channels
    geom = conv1d.__dict__.get("_xmc_geom")
    if geom is None:
        geom = conv1d.__dict__["_xmc_geom"] = ops.ConvGeom(E, n, 1, 1, 0)
    y = ops.linear(words_embs.reshape(B * T, E), conv1d.weight.view(n, E), conv1d.bias, geom, out_dtype=torch.float32)
    return y[:, :n].reshape(B, T, n)


def _head(global_cond, ctx, gconv):
    """grouped 1x1 head on cat(global condition, context) per concept, without the concatenation -> [B, C * p]"""
    return ops.grouped_vec(global_cond, ctx, gconv.weight, gconv.bias, gconv.groups).reshape(global_cond.size(0), -1)


class OutConceptBlock(nn.Module):
    def __init__(self, in_dim, cardinality, bottleneck_width, state_dim, text_dim, cond_dim, upsample, normalize=False):
        super(OutConceptBlock, self).__init__()
        self.cardinality, self.normalize, self.upsample = cardinality, normalize, upsample
        gw = cardinality * bottleneck_width
        cgw = cardinality * (cond_dim + state_dim)
        self.split_conv = HipConv2d(in_dim, gw, 1, 1, 0, bias=False)
        self.trans_gconv = _GroupedConv(gw, gw, 3, 1, groups=cardinality)
        if normalize:
            self.gn = nn.GroupNorm(cardinality, gw)
        self.concept_sampler1 = ConceptSampler(cardinality, bottleneck_width, state_dim, normalize=normalize)
        self.concept_reasoner1 = ConceptReasoner(cardinality, state_dim, normalize=normalize)
        self.word_conv1 = nn.Conv1d(text_dim, state_dim, 1, 1, 0, bias=False)
        self.concept_sampler2 = ConceptSampler(cardinality, bottleneck_width, state_dim, normalize=normalize)
        self.concept_reasoner2 = ConceptReasoner(cardinality, state_dim, normalize=normalize)
        self.word_conv2 = nn.Conv1d(text_dim, state_dim, 1, 1, 0, bias=False)
        self.gamma1_gconv = nn.Conv2d(cgw, gw, 1, 1, 0, groups=cardinality)
        self.beta1_gconv = nn.Conv2d(cgw, gw, 1, 1, 0, groups=cardinality)
        self.gamma2_gconv = nn.Conv2d(cgw, gw, 1, 1, 0, groups=cardinality)
        self.beta2_gconv = nn.Conv2d(cgw, gw, 1, 1, 0, groups=cardinality)

    def forward(self, x, global_cond, words_embs, mask):
        """x NHWC [B,h,w,Cin]; global_cond f32 [B,gc]; words_embs f32 [B,T,nef]; mask [B,T].  Returns the block output at the
        INPUT resolution (the caller applies the x2 upsample after the 1x1 output conv, see the module docstring)."""
        B = x.size(0)
        e = self.split_conv(x, act=ACT_RELU)
        e = self.trans_gconv(e)
        e = ops.groupnorm(e, self.gn.weight, self.gn.bias, self.cardinality, slope=0.0) if self.normalize else ops.lrelu(e, 0.0)
        st = self.concept_reasoner1(self.concept_sampler1(e))                                    # [B,C,p']
        ctx = _word_context(st, _project_words(words_embs, self.word_conv1), mask)
        g1, b1 = _head(global_cond, ctx, self.gamma1_gconv), _head(global_cond, ctx, self.beta1_gconv)
        if self.normalize and self.concept_reasoner2.training:
            # upstream's discarded call (432): only the BatchNorm1d running statistics remain
            r2 = self.concept_reasoner2
            ops.reasoner_stats_only(ctx, r2.proj_edge.weight, r2.bn)
        ctx2 = _word_context(ctx, _project_words(words_embs, self.word_conv2), mask)
        g2, b2 = _head(global_cond, ctx2, self.gamma2_gconv), _head(global_cond, ctx2, self.beta2_gconv)
        # relu(g2 * up(relu(g1*e+b1)) + b2) == up(relu(g2 * relu(g1*e+b1) + b2)): one two-stage pass at low resolution
        return ops.Affine2LreluFn.apply(e, g1, b1, g2, b2, 0.0)


class OCAttnResBlockUp(nn.Module):
    def __init__(self, in_dim, out_dim, gc_dim, text_dim, upsample, cardinality, bottleneck_width, normalize=True):
        super(OCAttnResBlockUp, self).__init__()
        self.learnable_sc = (in_dim != out_dim)
        self.normalize, self.upsample, self.cardinality = normalize, upsample, cardinality
        state_dim = 4
        gw = cardinality * bottleneck_width
        self.concept1 = OutConceptBlock(in_dim=in_dim, cardinality=cardinality, bottleneck_width=bottleneck_width,
                                        state_dim=state_dim, text_dim=text_dim, cond_dim=gc_dim, upsample=upsample,
                                        normalize=normalize)
        self.conv_out1 = HipConv2d(gw, out_dim, 1, 1, 0)
        if self.learnable_sc:
            self.c_sc = HipConv2d(in_dim, out_dim, 1, stride=1, padding=0)

    def forward(self, x, global_cond, words_embs, mask):
        r = self.conv_out1(self.concept1(x, global_cond, words_embs, mask))
        out = ops.axpby(self.c_sc(x) if self.learnable_sc else x, r, _one(x.device))
        return ops.upsample2(out) if self.upsample else out


class OutNetG(nn.Module):
    _attn_block = None          # the attention stage's class: OCAttnResBlockUp here, ICAttnResBlockUp in InNetG

    def __init__(self, cfg, **kwargs):
        super(OutNetG, self).__init__()
        attn_block = self._attn_block or OCAttnResBlockUp
        self.ngf = cfg.TRAIN.NCH
        noise_dim, nef = cfg.TRAIN.NOISE_DIM, cfg.TRAIN.NEF
        arch = gen_arch(img_size=cfg.IMG.SIZE, nch=self.ngf)
        c0 = arch['in_channels'][0]
        self.proj_sent = HipLinear(cfg.TEXT.EMBEDDING_DIM, nef)
        self.proj_word = nn.Conv1d(cfg.TEXT.EMBEDDING_DIM, nef, 1, 1, 0)      # parameter holder; runs as one GEMM over B*T words
        self._word_geom = ops.ConvGeom(cfg.TEXT.EMBEDDING_DIM, nef, 1, 1, 0)
        self.proj_cond = _Pad8Linear(noise_dim + nef, c0 * 4 * 4, row_perm=nhwc_feature_perm(c0))
        self.upblocks = nn.ModuleList(
            [ResBlockUp(in_dim=arch['in_channels'][i], out_dim=arch['out_channels'][i], cond_dim=noise_dim + nef,
                        upsample=arch['upsample'], normalize=cfg.GEN.NORMALIZE) for i in range(2)] +        # the LIST: truthy (262)
            [attn_block(in_dim=arch['in_channels'][i], out_dim=arch['out_channels'][i], gc_dim=noise_dim + nef,
                        text_dim=nef, upsample=arch['upsample'][i], cardinality=16, bottleneck_width=8,
                        normalize=cfg.GEN.NORMALIZE) for i in range(2, arch['depth'])])
        self.conv_out = nn.Sequential(
            nn.LeakyReLU(0.2, inplace=True),
            HipConv2d(arch['out_channels'][-1], 3, 3, 1, 1),
            nn.Tanh(),
        )

    def forward(self, noise, sent_embs, words_embs, mask):
        """noise [B,noise_dim], sent_embs [B,E], words_embs [B,E,T], mask [B,T] bool (True = padding) -> [B,3,S,S] f32."""
        B, E, T = words_embs.shape
        sent = self.proj_sent(sent_embs.float())
        w = ops.linear(words_embs.float().transpose(1, 2).reshape(B * T, E), self.proj_word.weight.view(-1, E),
                       self.proj_word.bias, self._word_geom, out_dtype=torch.float32)
        words = w[:, : self.proj_word.out_channels].reshape(B, T, -1)                             # [B,T,nef]
        global_cond = torch.cat([noise.float(), sent], dim=1)
        out = self.proj_cond(global_cond, out_dtype=ops.act_dtype()).view(B, 4, 4, -1)
        for gblock in self.upblocks:
            out = gblock(out, global_cond=global_cond, words_embs=words, mask=mask)
        out = self.conv_out[1](ops.lrelu(out), act=ACT_TANH)
        return ops.to_nchw(out, 3)


class CondConceptSampler(nn.Module):
    """Word-region sampler (concept_gan.py:516-580): query = grouped 1x1 of the map [-> GroupNorm], key = grouped Conv1d of the words
    repeated per concept [-> GroupNorm over (p', T)], cosine attention of every region over the unmasked words, mean over regions.
    ``cond_dim`` is the per-concept channel count of the words it is FED (repair 1 of the module docstring)."""

    def __init__(self, cardinality, bottleneck_width, state_dim, cond_dim, normalize=True):
        super(CondConceptSampler, self).__init__()
        self.cardinality, self.normalize, self.state_dim = cardinality, normalize, state_dim
        gw, sw = cardinality * bottleneck_width, cardinality * state_dim
        self.query_gconv = _GroupedConv(gw, sw, 1, 0)
        self.key_gconv = nn.Conv1d(cardinality * cond_dim, sw, 1, 1, 0, groups=cardinality, bias=False)
        if normalize:
            self.gn1 = nn.GroupNorm(cardinality, sw)
            self.gn2 = nn.GroupNorm(cardinality, sw)

    def forward(self, x, words_embs, mask):
        """x NHWC [B,h,w,C*p]; words_embs f32 [B,T,nef]; mask [B,T] (True = padding) -> context f32 [B,C,p']."""
        B, T, E = words_embs.shape
        C, P = self.cardinality, self.state_dim
        q = self.query_gconv(x)
        if self.normalize:
            q = ops.groupnorm(q, self.gn1.weight, self.gn1.bias, C, eps=self.gn1.eps)
        # the keys: every group sees the same words -- one GEMM over the B * T words, then GroupNorm over (p', T) per concept and the L2
        # normalisation over p' in one launch (csrc/concept_word.hip) -> [B,C,T,p']
        kraw = _project_words(words_embs, self.key_gconv)
        kh = ops.word_keys(kraw, self.gn2.weight if self.normalize else None, self.gn2.bias if self.normalize else None,
                           self.gn2.eps if self.normalize else 1e-5)
        return ops.word_region_pool(q, kh, mask)


class InConceptBlock(nn.Module):
    def __init__(self, in_dim, cardinality, bottleneck_width, state_dim, cond_dim, text_dim, normalize=False):
        super(InConceptBlock, self).__init__()
        self.cardinality, self.normalize = cardinality, normalize
        gw = cardinality * bottleneck_width
        cgw = cardinality * (cond_dim + state_dim)
        self.split_conv = HipConv2d(in_dim, gw, 1, 1, 0, bias=False)
        self.trans_gconv = _GroupedConv(gw, gw, 3, 1, groups=cardinality)
        if normalize:
            self.gn = nn.GroupNorm(cardinality, gw)
        self.concept_sampler1 = CondConceptSampler(cardinality, bottleneck_width, state_dim, text_dim, normalize=normalize)
        self.concept_reasoner1 = ConceptReasoner(cardinality, state_dim, normalize=normalize)
        self.concept_sampler2 = CondConceptSampler(cardinality, bottleneck_width, state_dim, text_dim, normalize=normalize)
        self.concept_reasoner2 = ConceptReasoner(cardinality, state_dim, normalize=normalize)
        self.gamma1_gconv = nn.Conv2d(cgw, gw, 1, 1, 0, groups=cardinality)
        self.beta1_gconv = nn.Conv2d(cgw, gw, 1, 1, 0, groups=cardinality)
        self.gamma2_gconv = nn.Conv2d(cgw, gw, 1, 1, 0, groups=cardinality)
        self.beta2_gconv = nn.Conv2d(cgw, gw, 1, 1, 0, groups=cardinality)

    def forward(self, x, global_cond, words_embs, mask):
        """x NHWC [B,h,w,Cin]; global_cond f32 [B,gc]; words_embs f32 [B,T,nef]; mask [B,T].  Returns the block output at the INPUT
        resolution (the caller upsamples once after the 1x1 output conv, see the module docstring)."""
        B = x.size(0)
        e = self.split_conv(x, act=ACT_RELU)
        e = self.trans_gconv(e)
        out = ops.groupnorm(e, self.gn.weight, self.gn.bias, self.cardinality, slope=0.0) if self.normalize else ops.lrelu(e, 0.0)
        for samp, reas, gm, bm in ((self.concept_sampler1, self.concept_reasoner1, self.gamma1_gconv, self.beta1_gconv),
                                   (self.concept_sampler2, self.concept_reasoner2, self.gamma2_gconv, self.beta2_gconv)):
            ctx = reas(samp(out, words_embs, mask))                                                # [B,C,p']
            out = ops.affine_act(out, _head(global_cond, ctx, gm), _head(global_cond, ctx, bm), 0.0)
        return out


class ICAttnResBlockUp(nn.Module):
    def __init__(self, in_dim, out_dim, gc_dim, text_dim, upsample, cardinality, bottleneck_width, normalize=True):
        super(ICAttnResBlockUp, self).__init__()
        self.learnable_sc = (in_dim != out_dim)
        self.normalize, self.upsample, self.cardinality = normalize, upsample, cardinality
        state_dim = 4
        gw = cardinality * bottleneck_width
        self.concept1 = InConceptBlock(in_dim=in_dim, cardinality=cardinality, bottleneck_width=bottleneck_width,
                                       state_dim=state_dim, cond_dim=gc_dim, text_dim=text_dim, normalize=normalize)
        self.conv_out1 = HipConv2d(gw, out_dim, 1, 1, 0)
        if self.learnable_sc:
            self.c_sc = HipConv2d(in_dim, out_dim, 1, stride=1, padding=0)

    def forward(self, x, global_cond, words_embs, mask):
        r = self.conv_out1(self.concept1(x, global_cond, words_embs, mask))
        out = ops.axpby(self.c_sc(x) if self.learnable_sc else x, r, _one(x.device))
        return ops.upsample2(out) if self.upsample else out


class InNetG(OutNetG):
    """The word-REGION attention generator (concept_gan.py:67-121), repaired as the module docstring states; stem, first two blocks,
    tail and forward are OutNetG's (upstream's two classes share them line for line: 69-103 / 246-279, 105-121 / 281-298)."""
    _attn_block = ICAttnResBlockUp
