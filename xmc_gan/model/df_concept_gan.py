"""Attention-modulation ("concept") generators with the reference's module API (model/df_concept_gan.py):
``InNetG`` (CONCEPT_IN_DF_GEN: sentence-conditioned region attention) and ``OutNetG`` (CONCEPT_OUT_DF_GEN:
self region attention + sentence->concept attention), plus the ``NetD`` stub that raises like upstream.

Same class names, constructor/forward signatures and ``state_dict()`` keys.  Heavy per-pixel work runs in HIP kernels:
1x1 / 3x3 / block-diagonal grouped convolutions on the MFMA implicit-GEMM kernels, GroupNorm(+LeakyReLU), the region
attention (scores, softmax over H*W, weighted sum), the per-sample channel modulation and -- for the sentence-conditioned
and self-attention blocks -- the per-sample concept algebra on ``[B,16,<=260]`` numbers (csrc/concept.hip).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from xmc_gan_amd import ops
from xmc_gan_amd.lib import ACT_LRELU, ACT_NONE

from .df_gan import NetG as _DFNetG, gen_arch, disc_arch, nhwc_feature_perm  # noqa: F401 (tables are shared upstream too)
from .modules import HipConv2d, HipLinear

CARD, PW, SD = 16, 8, 4          # cardinality, bottleneck width, state dim (hard-coded upstream: 110,118)


class _GroupedConv(nn.Conv2d):
    """nn.Conv2d(groups=16) parameter holder.  The kernels work on the block-diagonal expansion of the weight, written by the
    weight-pack kernel (cached per parameter version) and read back, diagonal blocks only, by the gradient-unpack kernel
    (ops.ConvGeom.groups)."""

    def __init__(self, in_dim, out_dim, k, pad, groups=CARD, bias=False):
        super().__init__(in_dim, out_dim, k, 1, pad, groups=groups, bias=bias)
        self.geom = ops.ConvGeom(in_dim, out_dim, k, 1, pad, groups=groups)

    def forward(self, x):
        return ops.conv2d(x, self.weight, self.bias, self.geom)


def _grouped_vec(x, conv):
    """grouped 1x1 conv applied to a per-sample vector laid out [B, groups, in_per_group] -> [B, groups, out_per_group]."""
    g = conv.groups
    w = conv.weight.view(g, conv.out_channels // g, conv.in_channels // g)
    y = torch.einsum('bgi,goi->bgo', x, w)
    if conv.bias is not None:
        y = y + conv.bias.view(1, g, -1)
    return y


class _ModMLP(nn.Sequential):
    """gamma/beta generator: grouped 1x1 -> LeakyReLU -> grouped 1x1 on the per-concept condition (178-200, 443-465)."""

    def __init__(self, cond_group_width):
        super().__init__(
            nn.Conv2d(cond_group_width, 2 * CARD * SD, 1, 1, 0, groups=CARD),
            nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(2 * CARD * SD, CARD * PW, 1, 1, 0, groups=CARD))

    def forward(self, cond):
        raise RuntimeError("_ModMLP holds the parameters of a gamma / beta head; it is evaluated inside ops.concept_stage")


class ConceptReasoner(nn.Module):
    def __init__(self, cardinality, state_dim, spec_norm=False, normalize=True):
        super(ConceptReasoner, self).__init__()
        self.cardinality = cardinality
        self.normalize = False                    # forced off upstream (308)
        self.proj_edge = nn.Linear(state_dim, cardinality, bias=False)

    def forward(self, x, **kwargs):
        raise RuntimeError("ConceptReasoner holds proj_edge; the reasoning step runs inside ops.concept_stage (csrc/concept.hip)")


class _SamplerBase(nn.Module):
    def _init_common(self, cardinality, bottleneck_width, state_dim, normalize):
        self.cardinality, self.normalize = cardinality, normalize
        gw, sw = cardinality * bottleneck_width, cardinality * state_dim
        self.key_gconv = _GroupedConv(gw, sw, 1, 0)
        self.value_gconv = nn.Conv2d(gw, sw, 1, 1, 0, groups=cardinality, bias=False)
        if normalize:
            self.gn1 = nn.GroupNorm(cardinality, sw)
            self.gn2 = nn.GroupNorm(cardinality, sw)

    # pool / _attend / ConceptSampler.forward: the un-fused form, used by the word-attention generator (concept_gan.py), whose
    # per-concept algebra is not in csrc/concept.hip
    def pool(self, x, q, scale):
        """region attention with an already normalised query: x [B,H,W,128], q [B,16,4] f32 -> pooled x [B,16,8]."""
        key = self.key_gconv(x)
        if self.normalize:
            key = ops.groupnorm(key, self.gn2.weight, self.gn2.bias, self.cardinality)
        return ops.attn_pool(key, q, x, self.cardinality, scale)

    def stage(self, x, q, scale, sent, head_params):
        """key projection [+ GroupNorm], region attention, concept head and the channel modulation lrelu(gamma * x + beta) as one
        autograd node (ops.ConceptStageFn): x [B,H,W,128], q [B,16,4] f32 (already normalised) -> [B,H,W,128].
        (the heads' sentence products have usually been computed for all stages at once: _ConceptNetG._hoist)"""
        return ops.concept_stage(x, q, sent, self.key_gconv.weight, self.gn2.weight if self.normalize else None,
                                 self.gn2.bias if self.normalize else None, self.key_gconv.geom, self.cardinality, scale,
                                 head_params, eps=self.gn2.eps if self.normalize else 1e-5, **self.__dict__.pop("_a_hoisted", {}))

    def _attend(self, x, q, scale):
        """x [B,H,W,128], q [B,16,4] f32 (already normalised) -> value-projected context [B,16,4]."""
        return ops.grouped_vec(None, self.pool(x, q, scale), self.value_gconv.weight, None, self.cardinality)


class CondConceptSampler(_SamplerBase):
    """sentence query -> region keys (reference 256-302)."""

    def __init__(self, cardinality, bottleneck_width, state_dim, cond_dim, normalize=True, spec_norm=False):
        super(CondConceptSampler, self).__init__()
        self.query_gconv = nn.Conv2d(cardinality * cond_dim, cardinality * state_dim, 1, 1, 0, groups=cardinality, bias=False)
        self._init_common(cardinality, bottleneck_width, state_dim, normalize)

    def forward(self, x, sent_embs):
        raise RuntimeError("CondConceptSampler is evaluated by InConceptBlock.forward (ops.concept_query + ops.concept_stage)")


class ConceptSampler(_SamplerBase):
    """global-average query -> region keys, scores scaled by rsqrt(state_dim) (reference 535-581)."""

    def __init__(self, cardinality, bottleneck_width, state_dim, spec_norm=False, normalize=True):
        super(ConceptSampler, self).__init__()
        self.query_gconv = nn.Conv2d(cardinality * bottleneck_width, cardinality * state_dim, 1, 1, 0, groups=cardinality, bias=False)
        self._init_common(cardinality, bottleneck_width, state_dim, normalize)
        self.register_buffer('norm', torch.rsqrt(torch.as_tensor(state_dim, dtype=torch.float)))
        self._scale = float(state_dim) ** -0.5

    def forward(self, x, **kwargs):
        # (the un-fused form, used by the word-attention generator: query = grouped 1x1 of the global average [+ GroupNorm] in one
        # launch, region attention, value projection -- csrc/concept.hip, concept_word.hip)
        q = ops.concept_gquery(ops.global_avgpool(x), self.query_gconv.weight, self.gn1.weight if self.normalize else None,
                               self.gn1.bias if self.normalize else None, self.gn1.eps if self.normalize else 1e-5)
        return self._attend(x, q, self._scale)


class _ConceptBlockBase(nn.Module):
    def _init_trunk(self, in_dim, cardinality, bottleneck_width, normalize):
        self.cardinality, self.normalize = cardinality, normalize
        gw = cardinality * bottleneck_width
        self.split_conv = HipConv2d(in_dim, gw, 1, 1, 0, bias=False)
        self.trans_gconv = _GroupedConv(gw, gw, 3, 1)
        if normalize:
            self.gn = nn.GroupNorm(cardinality, gw)

    def _trunk(self, x):
        e = self.split_conv(x, act=ACT_LRELU)
        e = self.trans_gconv(e)
        if self.normalize:
            return ops.groupnorm(e, self.gn.weight, self.gn.bias, self.cardinality, slope=0.2)
        return ops.lrelu(e)



class InConceptBlock(_ConceptBlockBase):
    def __init__(self, in_dim, cardinality, bottleneck_width, state_dim, cond_dim, normalize=False):
        super(InConceptBlock, self).__init__()
        self._init_trunk(in_dim, cardinality, bottleneck_width, normalize)
        cgw = cardinality * (state_dim + cond_dim)
        self.concept_sampler1 = CondConceptSampler(cardinality, bottleneck_width, state_dim, cond_dim, normalize)
        self.concept_reasoner1 = ConceptReasoner(cardinality, state_dim, normalize=normalize)
        self.concept_sampler2 = CondConceptSampler(cardinality, bottleneck_width, state_dim, cond_dim, normalize)
        self.concept_reasoner2 = ConceptReasoner(cardinality, state_dim, normalize=normalize)
        self.gamma1_gconv, self.beta1_gconv = _ModMLP(cgw), _ModMLP(cgw)
        self.gamma2_gconv, self.beta2_gconv = _ModMLP(cgw), _ModMLP(cgw)

    def forward(self, x, sent_embs):
        B = x.size(0)
        out = self._trunk(x)
        sent = sent_embs.float()
        for samp, reas, gm, bm in ((self.concept_sampler1, self.concept_reasoner1, self.gamma1_gconv, self.beta1_gconv),
                                   (self.concept_sampler2, self.concept_reasoner2, self.gamma2_gconv, self.beta2_gconv)):
            if not x.is_cuda:
                raise RuntimeError("InConceptBlock runs on the GPU only (no CPU fallback)")
            # per-sample concept algebra in two kernels per stage (csrc/concept.hip) around the region attention:
            #   sentence query + GroupNorm                                   (273-286)
            #   value projection, ConceptReasoner, gamma / beta grouped MLPs (238-253, 291-326)
            # (the generator's forward has usually computed every stage's query in one launch already: _ConceptNetG.forward)
            q = samp.__dict__.pop("_q_hoisted", None)
            if q is None:
                q = ops.concept_query(sent, samp.query_gconv.weight, samp.gn1.weight if samp.normalize else None,
                                      samp.gn1.bias if samp.normalize else None)
            out = samp.stage(out, q, 1.0, sent, (
                samp.value_gconv.weight, reas.proj_edge.weight,
                gm[0].weight, gm[0].bias, gm[2].weight, gm[2].bias, bm[0].weight, bm[0].bias, bm[2].weight, bm[2].bias))
        return out


class OutConceptBlock(_ConceptBlockBase):
    def __init__(self, in_dim, cardinality, bottleneck_width, state_dim, cond_dim, normalize=False):
        super(OutConceptBlock, self).__init__()
        self._init_trunk(in_dim, cardinality, bottleneck_width, normalize)
        cgw = cardinality * (state_dim + cond_dim)
        self.concept_sampler1 = ConceptSampler(cardinality, bottleneck_width, state_dim, normalize=normalize)
        self.concept_reasoner1 = ConceptReasoner(cardinality, state_dim, normalize=normalize)
        self.concept_sampler2 = ConceptSampler(cardinality, bottleneck_width, state_dim, normalize=normalize)
        self.concept_reasoner2 = ConceptReasoner(cardinality, state_dim, normalize=normalize)
        self.sent_linear1 = nn.Linear(cond_dim, state_dim, bias=False)
        self.sent_linear2 = nn.Linear(cond_dim, state_dim, bias=False)
        self.gamma1_gconv, self.beta1_gconv = _ModMLP(cgw), _ModMLP(cgw)
        self.gamma2_gconv, self.beta2_gconv = _ModMLP(cgw), _ModMLP(cgw)

    def forward(self, x, sent_embs):
        B = x.size(0)
        out = self._trunk(x)
        sent = sent_embs.float()
        if not x.is_cuda:
            raise RuntimeError("OutConceptBlock runs on the GPU only (no CPU fallback)")
        for samp, reas, sl, gm, bm in (
                (self.concept_sampler1, self.concept_reasoner1, self.sent_linear1, self.gamma1_gconv, self.beta1_gconv),
                (self.concept_sampler2, self.concept_reasoner2, self.sent_linear2, self.gamma2_gconv, self.beta2_gconv)):
            # the per-sample concept algebra in csrc/concept.hip, around the region attention:
            #   query from the globally averaged map + GroupNorm                                               (555-569)
            #   value projection, ConceptReasoner, sentence->concept softmax re-weighting, gamma / beta MLPs  (443-478, 570-581)
            q0 = ops.global_avgpool(out).view(B, -1)
            q = ops.concept_gquery(q0, samp.query_gconv.weight, samp.gn1.weight if samp.normalize else None,
                                   samp.gn1.bias if samp.normalize else None)
            out = samp.stage(out, q, samp._scale, sent, (
                samp.value_gconv.weight, reas.proj_edge.weight,
                gm[0].weight, gm[0].bias, gm[2].weight, gm[2].bias, bm[0].weight, bm[0].bias, bm[2].weight, bm[2].bias,
                sl.weight))
        return out


class _AttnGBlock(nn.Module):
    def _init(self, block_cls, in_dim, out_dim, cond_dim, upsample, cardinality, bottleneck_width, normalize, k):
        self.learnable_sc = (in_dim != out_dim)
        self.upsample, self.cardinality = upsample, cardinality
        gw = cardinality * bottleneck_width
        self.concept1 = block_cls(in_dim, cardinality, bottleneck_width, 4, cond_dim, normalize)
        self.concept2 = block_cls(out_dim, cardinality, bottleneck_width, 4, cond_dim, normalize)
        self.conv_out1 = HipConv2d(gw, out_dim, k, 1, k // 2)
        self.conv_out2 = HipConv2d(gw, out_dim, k, 1, k // 2)
        self.gamma = nn.Parameter(torch.zeros(1))
        if self.learnable_sc:
            self.c_sc = HipConv2d(in_dim, out_dim, 1, stride=1, padding=0)

    def shortcut(self, x):
        return self.c_sc(x) if self.learnable_sc else x

    def residual(self, x, sent_embs):
        out = self.conv_out1(self.concept1(x, sent_embs), act=ACT_LRELU)
        return self.conv_out2(self.concept2(out, sent_embs))

    def forward(self, x, sent_embs):
        out = ops.axpby(self.shortcut(x), self.residual(x, sent_embs), self.gamma)
        return ops.upsample2(out) if self.upsample else out


class ICAttnG_Block(_AttnGBlock):
    def __init__(self, in_dim, out_dim, cond_dim, upsample, cardinality=16, bottleneck_width=8, normalize=True):
        super(ICAttnG_Block, self).__init__()
        self._init(InConceptBlock, in_dim, out_dim, cond_dim, upsample, cardinality, bottleneck_width, normalize, 3)


class OCAG_Block(_AttnGBlock):
    def __init__(self, in_dim, out_dim, cond_dim, upsample, cardinality=16, bottleneck_width=8, normalize=True):
        super(OCAG_Block, self).__init__()
        self.normalize = normalize
        self._init(OutConceptBlock, in_dim, out_dim, cond_dim, upsample, cardinality, bottleneck_width, normalize, 1)


class _ConceptNetG(_DFNetG):
    """stem (proj_noise -> [B,8*ngf,4,4]) and tail (LeakyReLU, Conv3x3, Tanh) are DF_GEN's (65-105 / 328-367)."""
    _block = None

    def __init__(self, cfg, **kwargs):
        nn.Module.__init__(self)
        self.ngf = cfg.TRAIN.NCH
        arch = gen_arch(img_size=cfg.IMG.SIZE, nch=self.ngf)
        self.proj_noise = HipLinear(cfg.TRAIN.NOISE_DIM, (8 * self.ngf) * 16, row_perm=nhwc_feature_perm(8 * self.ngf))
        self.proj_sent = HipLinear(cfg.TEXT.EMBEDDING_DIM, cfg.TRAIN.NEF) \
            if (cfg.TEXT.EMBEDDING_DIM != cfg.TRAIN.NEF) else nn.Identity()
        self.upblocks = nn.ModuleList(
            [self._block(in_dim=arch['in_channels'][i], out_dim=arch['out_channels'][i], cond_dim=cfg.TRAIN.NEF,
                         upsample=arch['upsample'][i], normalize=cfg.GEN.NORMALIZE) for i in range(arch['depth'])])
        self.conv_out = nn.Sequential(nn.LeakyReLU(0.2, inplace=True), HipConv2d(arch['out_channels'][-1], 3, 3, 1, 1), nn.Tanh())

    def forward(self, noise, sent_embs, return_nhwc=False, nhwc_dst=None, **kwargs):
        sent_embs = self.proj_sent(sent_embs.float())
        out = self.stem(noise)
        self._hoist(sent_embs)
        for gblock in self.upblocks:
            out = gblock(out, sent_embs)
        return self.tail(out, False, return_nhwc, nhwc_dst)

    def _hoist(self, sent):
        """What every sampler stage computes from the sentence vector ALONE, once for the whole generator, handed to the stages through
        one-shot attributes: the sentence part of the gamma / beta heads' first layer (df_concept_gan.py:238-253; one grouped GEMM instead of
        a 256-row dot per sample in each of the 24-28 head launches) and, for the sentence-attention kind, the queries of all
        CondConceptSampler stages (273-286; ops.concept_query_all: one launch, backward one GroupNorm-backward launch and ONE batch product)."""
        if not sent.is_cuda:
            return
        sent = sent.float()
        blocks = [m for m in self.modules() if isinstance(m, (InConceptBlock, OutConceptBlock))]
        if blocks and not ops.debug_switch("no_head_hoist"):
            stages = [(m.concept_sampler1, m.gamma1_gconv, m.beta1_gconv) for m in blocks] + [(m.concept_sampler2, m.gamma2_gconv, m.beta2_gconv) for m in blocks]
            A, hoist = ops.head_sentence_products(sent, [(gm[0].weight, bm[0].weight) for _, gm, bm in stages])
            for k, (s_, _, _) in enumerate(stages):
                s_.__dict__["_a_hoisted"] = dict(a_pre=A[k], hoist=(hoist, k))
        samplers = [s_ for m in blocks if isinstance(m, InConceptBlock) for s_ in (m.concept_sampler1, m.concept_sampler2)]
        if not samplers or len(samplers) > 32 or ops.debug_switch("no_query_hoist"):
            return
        qs = ops.concept_query_all(sent, [(s_.query_gconv.weight, s_.gn1.weight if s_.normalize else None,
                                           s_.gn1.bias if s_.normalize else None) for s_ in samplers])
        for s_, q in zip(samplers, qs):
            s_.__dict__["_q_hoisted"] = q


class InNetG(_ConceptNetG):
    _block = ICAttnG_Block


class OutNetG(_ConceptNetG):
    _block = OCAG_Block


class NetD(nn.Module):
    def __init__(self, cfg, **kwargs):
        super(NetD, self).__init__()
        raise NotImplementedError      # as upstream (df_concept_gan.py:587)
