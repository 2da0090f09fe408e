"""CPU oracle for the XMC-GAN G+D training step  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A functional, single-file restatement (plain PyTorch, fp32, CPU) of the arithmetic the
reference performs on its hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file; the product (``xmc_gan/`` and
``xmc-gan_amd/``) never does and fails loudly when the HIP library is missing.

Parity status: PINNED.  ``oracle/make_golden.py`` runs the reference's own modules and its
real ``train()`` loop in the build container (recipe: SURVEY.md section 8c) and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function here against
those vectors.

Parameters are passed as plain ``dict[str, Tensor]`` keyed exactly like the reference's
``state_dict()``, so the same dict can be loaded into the reference, this
oracle and the product.  Every function cites the reference lines it restates
(paths relative to /root/reference/xmc_gan/).
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

LRELU = 0.2
CARD, PW, SD = 16, 8, 4  # cardinality, bottleneck width p, state dim p' (df_concept_gan.py:110,118)


# ----------------------------------------------------------------------------------------
# quantisation-aware mode (checks the engine's bf16 path)
# ----------------------------------------------------------------------------------------
# The reference computes in fp32 throughout.  The MI355X engine's default mode stores every activation tensor and the packed
# convolution weights in bf16 (f32 accumulation, f32 parameters / gradients of parameters / reductions).  With `quant(True)`
# the SAME restatement rounds to bf16 at exactly those storage points -- forward values where the engine writes an activation,
# gradient values where it writes a gradient tensor, convolution weights where it packs them (straight-through for the weight
# gradient, which the engine accumulates in f32) -- and keeps all arithmetic in f32.  What then remains between this oracle and
# the engine is summation order, i.e. kernel error proper, which separates it from the error of the number format.
# Covers DF_GEN + DF_DISC (the benched path) and, since round 4, the attention-modulation generators (CONCEPT_IN / CONCEPT_OUT_DF_GEN); the
# mode is off unless a test turns it on, the goldens never see it.
_QUANT = False
_QSKIP = frozenset()      # storage sites that stay f32 although the mode is on (the per-site ladder, tests/diag/quant_ladder.py)
_QGRAD = True             # round the gradient that flows back through a storage site as well

# the storage sites of the engine's bf16 mode, by tag (`q(x, site)` / `qw(w, site)`):
#   d.img  the image the discriminator reads          d.conv_img, d.r0, d.r2  its convolution outputs
#   d.pool the pooled shortcut input                  d.sc  the 1x1 shortcut output        d.sum  the block sum
#   d.w    packed discriminator weights               h.c / h.m / h.w  COND_DNET: condition, joint_conv.0 output, weights
#   g.stem, g.aff (affine-affine-LeakyReLU passes), g.c1, g.c2, g.sc, g.sum, g.act (tail LeakyReLU), g.img, g.w
#   d.last the LAST block sum (the [B,16*NCH,4,4] feature map COND_DNET reads): a site of its own only when it is skipped -- the
#          engine's "f32 head" option keeps that map, the condition, joint_conv.0's output and the head's weights in f32
QUANT_SITES = ("d.img", "d.conv_img", "d.r0", "d.r2", "d.pool", "d.sc", "d.sum", "d.last", "d.w", "h.c", "h.m", "h.w",
               "g.stem", "g.aff", "g.c1", "g.c2", "g.sc", "g.sum", "g.act", "g.img", "g.w",
               "g.c.split", "g.c.trans", "g.c.trunk", "g.c.key", "g.c.keyn", "g.c.mod", "g.c.out1", "g.c.out2")
#   g.c.*  the attention-modulation generators (round 4): split_conv / trans_gconv outputs, the trunk after GroupNorm + LeakyReLU, the
#          key projection and its GroupNorm, the modulated map of a concept stage, conv_out1 / conv_out2 outputs.  Everything per (sample,
#          concept) -- queries, attention statistics, pooled contexts, the heads' gamma / beta -- is f32 in the engine


class quant:
    """context manager / switch: `with X.quant(True): ...`.  ``skip``: site tags (prefix match, e.g. "d." or "g.w") that keep
    f32 storage; ``grad=False``: forward values are rounded, gradients pass through unrounded."""

    def __init__(self, on=True, skip=(), grad=True, fmt=torch.bfloat16, grad_scale=None, only=None, precise=None):
        """``precise`` (round 5; default: on for IEEE half, off for bf16 -- the engine's defaults, ops.precise_trunk): the discriminator's
        PRECISE TRUNK.  The generator's learned shortcuts c_sc: weights exact.  The discriminator, outside the pass whose backward is
        differentiated again: the learned shortcuts' weights exact (a 16-bit hi + lo
        pair or exact-f32 MFMA in the engine; the composed stem's shortcut stays rounded); on maps of <= 8x8 pixels the shortcut, the
        block sum (its branch unrounded, f32 destination) and the pooled by-product stay f32 -- the next block's conv_r[0] reads the
        rounded sum --; COND_DNET runs in f32 on the f32 map of the last block.  Gradients are rounded where they were.
        ``only`` (round 5, the per-layer ladder): round NOTHING but these sites; an entry "site@layer" (e.g. "d.w@b3.s", "d.sum@b2")
        names one layer of a discriminator site -- `skip` takes such entries too (that layer alone keeps f32).
        ``grad_scale``: the factor the engine's backward passes run at in this format (ops.loss_scale(): 4096 for IEEE half, whose
        5 exponent bits would otherwise put 1/B-sized gradients among the denormals; 1 for bf16) -- a gradient tensor is rounded as
        round(g * scale) / scale, which is what the engine stores and later divides out."""
        self.on, self.skip, self.grad, self.fmt = bool(on), tuple(skip), bool(grad), fmt
        self.only = None if only is None else tuple(only)
        self.precise = (fmt == torch.float16) if precise is None else bool(precise)
        self.gscale = float(grad_scale) if grad_scale is not None else (4096.0 if fmt == torch.float16 else 1.0)

    def __enter__(self):
        global _QUANT, _QSKIP, _QGRAD, _QFMT, _QGSCALE, _QLAYER, _QPRECISE
        self.prev = (_QUANT, _QSKIP, _QGRAD, _QFMT, _QGSCALE, _QLAYER, _QPRECISE)
        _QUANT, _QGRAD, _QFMT, _QGSCALE, _QPRECISE = self.on, self.grad, self.fmt, self.gscale, self.precise and self.on
        whole = tuple(p_ for p_ in self.skip if "@" not in p_)
        _QSKIP = frozenset(t for t in QUANT_SITES if any(t.startswith(p_) for p_ in whole))
        _QLAYER = {p_: False for p_ in self.skip if "@" in p_}                 # "site@layer" -> rounded?
        if self.only is not None:
            whole = tuple(p_ for p_ in self.only if "@" not in p_)
            per_layer = {p_.split("@")[0] for p_ in self.only if "@" in p_}
            # ("d.conv_img" / "d.last" in _QSKIP also switch the composed stem off / split the last block sum off: not meant here)
            _QSKIP = frozenset(t for t in QUANT_SITES if not any(t.startswith(p_) for p_ in whole) and t not in per_layer
                               and t not in ("d.conv_img", "d.last"))
            _QLAYER = {p_: True for p_ in self.only if "@" in p_}
            _QLAYER.update({t + "@*": False for t in per_layer})                 # the site's other layers: f32
        return self

    def __exit__(self, *a):
        global _QUANT, _QSKIP, _QGRAD, _QFMT, _QGSCALE, _QLAYER, _QPRECISE
        _QUANT, _QSKIP, _QGRAD, _QFMT, _QGSCALE, _QLAYER, _QPRECISE = self.prev


_QPRECISE = False          # the discriminator's precise trunk (quant.__init__)
_QLAYER = {}               # per-layer overrides of the ladder: "site@layer" -> rounded?  ("site@*": the site's default)
_QFMT = torch.bfloat16     # the 16-bit storage format the mode rounds to (torch.float16: the what-if rung of the ladder)
_QGSCALE = 1.0             # loss scale of the backward passes (see quant.__init__)


def _bf16(t):
    return t.to(_QFMT).to(torch.float32)


def _bf16g(g):
    """a stored GRADIENT tensor: rounded at the engine's loss scale"""
    return _bf16(g * _QGSCALE) / _QGSCALE if _QGSCALE != 1.0 else _bf16(g)


class _QAct(torch.autograd.Function):
    """an activation tensor as the engine stores it: value and gradient both rounded to bf16"""

    @staticmethod
    def forward(ctx, x):
        return _bf16(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16g(g) if _QGRAD else g


class _QGradOnly(torch.autograd.Function):
    """a tensor the engine keeps in f32 whose GRADIENT it stores in the 16-bit format (the precise trunk: forward exact)"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16g(g) if _QGRAD else g


class _QWeight(torch.autograd.Function):
    """a packed convolution weight: rounded for the forward / data-gradient kernels, f32 weight gradient passed through"""

    @staticmethod
    def forward(ctx, w):
        return _bf16(w)

    @staticmethod
    def backward(ctx, g):
        return g


class _QBranchTimesGamma(torch.autograd.Function):
    """gamma * LeakyReLU(z) of a discriminator block as the engine's first-order path stores it (xmc_gan_amd/ops/_nodes_block.py ResDFn /
    ResDBwdFn): forward, the branch is rounded where it enters the block sum; backward, the engine keeps only the
    branch's SIGN, rounds s * dout (`xmc_signmask_apply`) and applies gamma in f32 in the epilogue of the data gradient that
    follows -- so the gradient handed to the convolution is gamma * round(s * dout), not round(gamma * dout) * s."""

    @staticmethod
    def forward(ctx, z, gamma, round_fwd=True):
        r = F.leaky_relu(z, LRELU)
        if round_fwd:                  # (False: the precise trunk's f32 block sum takes the branch from the accumulators)
            r = _bf16(r)
        ctx.save_for_backward(z, gamma, r)
        return gamma * r

    @staticmethod
    def backward(ctx, g):
        z, gamma, r = ctx.saved_tensors
        sg = torch.where(z > 0, g, LRELU * g)
        if _QGRAD:
            sg = _bf16g(sg)
        return gamma * sg, (g * r).sum().reshape(gamma.shape), None


def _rounded(site, layer):
    if not _QUANT or site in _QSKIP:
        return False
    if _QLAYER and layer is not None:
        return _QLAYER.get(f"{site}@{layer}", _QLAYER.get(site + "@*", True))
    return _QLAYER.get(site + "@*", True) if _QLAYER else True


def q(x, site=None, layer=None, grad_only=False):
    if not _rounded(site, layer):
        return x
    return _QGradOnly.apply(x) if grad_only else _QAct.apply(x)


def qw(w, site=None, layer=None):
    return _QWeight.apply(w) if _rounded(site, layer) else w


# ----------------------------------------------------------------------------------------
# hyper-parameters
# ----------------------------------------------------------------------------------------
@dataclass
class Hyper:
    """The subset of the reference cfg (config/gan.py:7-90) the hot path reads."""
    img_size: int = 64
    nch: int = 32
    nef: int = 256
    noise_dim: int = 100
    text_dim: int = 256
    gen: str = "DF_GEN"           # cfg.GEN.ENCODER_NAME
    normalize: bool = True        # cfg.GEN.NORMALIZE
    img_match: bool = True        # cfg.DISC.IMG_MATCH
    sent_match: bool = False      # cfg.DISC.SENT_MATCH
    seperate: bool = False        # cfg.DISC.SEPERATE (sic)
    rmis: bool = True             # cfg.TRAIN.RMIS_LOSS
    magp: bool = False            # cfg.TRAIN.MAGP
    enc_sent: bool = True         # cfg.TRAIN.ENCODER_LOSS.SENT
    enc_disc: bool = True         # cfg.TRAIN.ENCODER_LOSS.DISC
    b_global: bool = False        # cfg.TRAIN.ENCODER_LOSS.B_GLOBAL
    smooth_mismatch: float = 1.0
    smooth_global: float = 0.0
    smooth_sent: float = 1.0
    smooth_disc: float = 1.0
    n_critic: int = 1
    spec_norm: bool = False       # cfg.DISC.SPEC_NORM (every shipped yml: False; config/gan.py:62 default True)
    g_lr: float = 1e-4
    g_betas: tuple = (0.0, 0.9)
    d_lr: float = 4e-4
    d_betas: tuple = (0.0, 0.9)

    @staticmethod
    def from_cfg(cfg) -> "Hyper":
        """cfg: nested mapping with the reference schema (attribute or item access)."""
        g = lambda node, k: node[k] if isinstance(node, dict) else getattr(node, k)
        T, D, E = g(cfg, "TRAIN"), g(cfg, "DISC"), g(g(cfg, "TRAIN"), "ENCODER_LOSS")
        S, O = g(T, "SMOOTH"), g(T, "OPT")
        return Hyper(
            img_size=g(g(cfg, "IMG"), "SIZE"), nch=g(T, "NCH"), nef=g(T, "NEF"),
            noise_dim=g(T, "NOISE_DIM"), text_dim=g(g(cfg, "TEXT"), "EMBEDDING_DIM"),
            gen=g(g(cfg, "GEN"), "ENCODER_NAME"), normalize=g(g(cfg, "GEN"), "NORMALIZE"),
            img_match=g(D, "IMG_MATCH"), sent_match=g(D, "SENT_MATCH"), seperate=g(D, "SEPERATE"),
            rmis=g(T, "RMIS_LOSS"), magp=g(T, "MAGP"), enc_sent=g(E, "SENT"), enc_disc=g(E, "DISC"),
            b_global=g(E, "B_GLOBAL"), smooth_mismatch=g(S, "MISMATCH"), smooth_global=g(S, "GLOBAL"),
            smooth_sent=g(S, "SENT"), smooth_disc=g(S, "DISC"), n_critic=g(T, "N_CRITIC"),
            spec_norm=bool(g(D, "SPEC_NORM")),
            g_lr=g(O, "G_LR"), g_betas=(g(O, "G_BETA1"), g(O, "G_BETA2")),
            d_lr=g(O, "D_LR"), d_betas=(g(O, "D_BETA1"), g(O, "D_BETA2")))


# ----------------------------------------------------------------------------------------
# architecture tables  (model/df_gan.py:9-61, duplicated at model/df_concept_gan.py:10-62)
# ----------------------------------------------------------------------------------------
def gen_arch(img_size: int, nch: int):
    assert img_size in (64, 128, 256)
    depth = {64: 5, 128: 6, 256: 7}[img_size]           # the last block keeps its resolution
    mult_in = [8] * (depth - 2) + [4, 2]                # 64: [8,8,8,4,2] ... 256: [8,8,8,8,8,4,2]
    mult_out = mult_in[1:] + [1]
    return dict(cin=[m * nch for m in mult_in], cout=[m * nch for m in mult_out],
                upsample=[True] * (depth - 1) + [False], depth=depth)


def disc_arch(img_size: int, nch: int):
    assert img_size in (64, 128, 256)
    depth = {64: 5, 128: 6, 256: 7}[img_size]
    mult_out = [1, 2, 4, 8, 16, 16, 16][:depth]
    cout = [m * nch for m in mult_out]
    cin = [3] + cout[:-1]
    return dict(cin=cin, cout=cout, depth=depth)


# ----------------------------------------------------------------------------------------
# state_dict key/shape tables (pinned against the reference's state_dict() in the golden test)
# ----------------------------------------------------------------------------------------
def _affine_shapes(prefix, nfeat, cond):
    s = {}
    for br in ("fc_gamma", "fc_beta"):
        s[f"{prefix}.{br}.linear1.weight"] = (256, cond)
        s[f"{prefix}.{br}.linear1.bias"] = (256,)
        s[f"{prefix}.{br}.linear2.weight"] = (nfeat, 256)
        s[f"{prefix}.{br}.linear2.bias"] = (nfeat,)
    return s


def _stem_tail_shapes(h: Hyper, arch):
    s = {"proj_noise.weight": (8 * h.nch * 16, h.noise_dim), "proj_noise.bias": (8 * h.nch * 16,)}
    if h.text_dim != h.nef:
        s["proj_sent.weight"] = (h.nef, h.text_dim)
        s["proj_sent.bias"] = (h.nef,)
    return s


def netg_shapes(h: Hyper):
    """DF_GEN state_dict (model/df_gan.py:64-103,179-263), in registration order."""
    a = gen_arch(h.img_size, h.nch)
    s = _stem_tail_shapes(h, a)
    for i in range(a["depth"]):
        ci, co, p = a["cin"][i], a["cout"][i], f"upblocks.{i}"
        s[f"{p}.gamma"] = (1,)
        s[f"{p}.c1.weight"] = (co, ci, 3, 3); s[f"{p}.c1.bias"] = (co,)
        s[f"{p}.c2.weight"] = (co, co, 3, 3); s[f"{p}.c2.bias"] = (co,)
        s.update(_affine_shapes(f"{p}.affine0", ci, h.nef))
        s.update(_affine_shapes(f"{p}.affine1", ci, h.nef))
        s.update(_affine_shapes(f"{p}.affine2", co, h.nef))
        s.update(_affine_shapes(f"{p}.affine3", co, h.nef))
        if ci != co:
            s[f"{p}.c_sc.weight"] = (co, ci, 1, 1); s[f"{p}.c_sc.bias"] = (co,)
    s["conv_out.1.weight"] = (3, a["cout"][-1], 3, 3); s["conv_out.1.bias"] = (3,)
    return s


def netd_shapes(h: Hyper):
    """DF_DISC state_dict (model/df_gan.py:106-176,266-294). conv_s always exists (280)."""
    a = disc_arch(h.img_size, h.nch)
    s = {"conv_img.weight": (a["cout"][0], 3, 3, 3), "conv_img.bias": (a["cout"][0],)}
    for i in range(1, a["depth"]):
        ci, co, p = a["cin"][i], a["cout"][i], f"downblocks.{i - 1}"
        s[f"{p}.gamma"] = (1,)
        s[f"{p}.conv_r.0.weight"] = (co, ci, 4, 4)
        s[f"{p}.conv_r.2.weight"] = (co, co, 3, 3)
        s[f"{p}.conv_s.weight"] = (co, ci, 1, 1); s[f"{p}.conv_s.bias"] = (co,)
    ndf16 = 16 * h.nch
    if h.img_match:                                        # df_gan.py:143-145
        s["COND_DNET.proj_match.weight"] = (h.nef, ndf16); s["COND_DNET.proj_match.bias"] = (h.nef,)
        cond = h.nef
    elif h.sent_match:                                     # 146-148
        s["COND_DNET.proj_match.weight"] = (ndf16, h.nef); s["COND_DNET.proj_match.bias"] = (ndf16,)
        cond = ndf16
    elif h.seperate and h.text_dim != h.nef:               # 149-151
        s["COND_DNET.proj_match.weight"] = (h.nef, h.text_dim); s["COND_DNET.proj_match.bias"] = (h.nef,)
        cond = h.nef
    else:                                                  # 152-154
        cond = h.text_dim
    s["COND_DNET.joint_conv.0.weight"] = (2 * h.nch, ndf16 + cond, 3, 3)
    s["COND_DNET.joint_conv.2.weight"] = (1, 2 * h.nch, 4, 4)
    if h.spec_norm:
        # torch.nn.utils.spectral_norm (modules.py:16-17,31-32): `weight` becomes the parameter `weight_orig` plus the
        # power-iteration buffers `weight_u` [out] and `weight_v` [in * kh * kw]
        sn = {}
        for k, shp in s.items():
            if k.endswith(".weight"):
                n_in = 1
                for d_ in shp[1:]:
                    n_in *= d_
                sn[k + "_orig"] = shp
                sn[k + "_u"] = (shp[0],)
                sn[k + "_v"] = (n_in,)
            else:
                sn[k] = shp
        s = sn
    return s


def _concept_block_shapes(p, in_dim, h: Hyper, kind):
    """InConceptBlock (df_concept_gan.py:159-200) / OutConceptBlock (421-465)."""
    gw, sw = CARD * PW, CARD * SD
    cgw = CARD * (SD + h.nef)
    s = {f"{p}.split_conv.weight": (gw, in_dim, 1, 1), f"{p}.trans_gconv.weight": (gw, PW, 3, 3)}
    if h.normalize:
        s[f"{p}.gn.weight"] = (gw,); s[f"{p}.gn.bias"] = (gw,)
    for j in (1, 2):
        q = f"{p}.concept_sampler{j}"
        if kind == "in":   # CondConceptSampler 256-271
            s[f"{q}.query_gconv.weight"] = (sw, h.nef, 1, 1)
        else:              # ConceptSampler 535-552 (registers buffer 'norm' last)
            s[f"{q}.query_gconv.weight"] = (sw, PW, 1, 1)
        s[f"{q}.key_gconv.weight"] = (sw, PW, 1, 1)
        s[f"{q}.value_gconv.weight"] = (sw, PW, 1, 1)
        if h.normalize:
            for gn in ("gn1", "gn2"):
                s[f"{q}.{gn}.weight"] = (sw,); s[f"{q}.{gn}.bias"] = (sw,)
        if kind == "out":
            s[f"{q}.norm"] = ()
        s[f"{p}.concept_reasoner{j}.proj_edge.weight"] = (CARD, SD)
        # registration order in the reference: sampler1, reasoner1, sampler2, reasoner2
    if kind == "out":
        s[f"{p}.sent_linear1.weight"] = (SD, h.nef)
        s[f"{p}.sent_linear2.weight"] = (SD, h.nef)
    for name in ("gamma1_gconv", "beta1_gconv", "gamma2_gconv", "beta2_gconv"):
        s[f"{p}.{name}.0.weight"] = (2 * sw, SD + h.nef, 1, 1); s[f"{p}.{name}.0.bias"] = (2 * sw,)
        s[f"{p}.{name}.2.weight"] = (gw, 2 * SD, 1, 1); s[f"{p}.{name}.2.bias"] = (gw,)
    return s


def concept_netg_shapes(h: Hyper):
    """CONCEPT_IN_DF_GEN / CONCEPT_OUT_DF_GEN state_dict (df_concept_gan.py:65-156, 328-418)."""
    kind = {"CONCEPT_IN_DF_GEN": "in", "CONCEPT_OUT_DF_GEN": "out"}[h.gen]
    a = gen_arch(h.img_size, h.nch)
    s = _stem_tail_shapes(h, a)
    gw = CARD * PW
    k = 3 if kind == "in" else 1      # conv_out1/2 are 3x3 in ICAttnG_Block (125-126), 1x1 in OCAG_Block (387-388)
    for i in range(a["depth"]):
        ci, co, p = a["cin"][i], a["cout"][i], f"upblocks.{i}"
        s[f"{p}.gamma"] = (1,)
        s.update(_concept_block_shapes(f"{p}.concept1", ci, h, kind))
        s.update(_concept_block_shapes(f"{p}.concept2", co, h, kind))
        s[f"{p}.conv_out1.weight"] = (co, gw, k, k); s[f"{p}.conv_out1.bias"] = (co,)
        s[f"{p}.conv_out2.weight"] = (co, gw, k, k); s[f"{p}.conv_out2.bias"] = (co,)
        if ci != co:
            s[f"{p}.c_sc.weight"] = (co, ci, 1, 1); s[f"{p}.c_sc.bias"] = (co,)
    s["conv_out.1.weight"] = (3, a["cout"][-1], 3, 3); s["conv_out.1.bias"] = (3,)
    return s


def word_gen_arch(img_size: int, nch: int):
    """channel tables of the word-attention generators (concept_gan.py:11-37)"""
    mult = {256: ([16, 16, 8, 8, 4, 2, 1], [16, 8, 8, 4, 2, 1, 1]),
            128: ([16, 8, 8, 4, 2, 1], [8, 8, 4, 2, 1, 1]),
            64: ([8, 8, 4, 2, 1], [8, 4, 2, 1, 1])}[img_size]
    depth = len(mult[0])
    return dict(in_channels=[m * nch for m in mult[0]], out_channels=[m * nch for m in mult[1]],
                upsample=[True] * (depth - 1) + [False], depth=depth)


def _bn_shapes(s, p, n):
    s[f"{p}.weight"] = (n,); s[f"{p}.bias"] = (n,)
    s[f"{p}.running_mean"] = (n,); s[f"{p}.running_var"] = (n,); s[f"{p}.num_batches_tracked"] = ()


def word_netg_shapes(h: Hyper):
    """state_dict of concept_gan.OutNetG (concept_gan.py:244-279): two BatchNorm-conditional ResBlockUp (454-476), then
    OCAttnResBlockUp (300-318) with the word-attention OutConceptBlock (346-371)."""
    a = word_gen_arch(h.img_size, h.nch)
    gc = h.noise_dim + h.nef
    gw, sw = CARD * PW, CARD * SD
    s = {"proj_sent.weight": (h.nef, h.text_dim), "proj_sent.bias": (h.nef,),
         "proj_word.weight": (h.nef, h.text_dim, 1), "proj_word.bias": (h.nef,),
         "proj_cond.weight": (a["in_channels"][0] * 16, gc), "proj_cond.bias": (a["in_channels"][0] * 16,)}
    for i in range(a["depth"]):
        p, ci, co = f"upblocks.{i}", a["in_channels"][i], a["out_channels"][i]
        if i < 2:
            s[f"{p}.c1.weight"] = (co, ci, 3, 3); s[f"{p}.c1.bias"] = (co,)
            s[f"{p}.c2.weight"] = (co, co, 3, 3); s[f"{p}.c2.bias"] = (co,)
            if h.normalize:
                _bn_shapes(s, f"{p}.bn1", ci)
                _bn_shapes(s, f"{p}.bn2", co)
            s[f"{p}.linear_gamma1.weight"] = (ci, gc); s[f"{p}.linear_beta1.weight"] = (ci, gc)
            s[f"{p}.linear_gamma2.weight"] = (co, gc); s[f"{p}.linaer_beta2.weight"] = (co, gc)     # sic (473)
        else:
            q = f"{p}.concept1"
            s[f"{q}.split_conv.weight"] = (gw, ci, 1, 1)
            s[f"{q}.trans_gconv.weight"] = (gw, PW, 3, 3)
            if h.normalize:
                s[f"{q}.gn.weight"] = (gw,); s[f"{q}.gn.bias"] = (gw,)
            for j in (1, 2):
                for nm in ("query", "key", "value"):
                    s[f"{q}.concept_sampler{j}.{nm}_gconv.weight"] = (sw, PW, 1, 1)
                if h.normalize:
                    for g_ in ("gn1", "gn2"):
                        s[f"{q}.concept_sampler{j}.{g_}.weight"] = (sw,); s[f"{q}.concept_sampler{j}.{g_}.bias"] = (sw,)
                s[f"{q}.concept_sampler{j}.norm"] = ()
                s[f"{q}.concept_reasoner{j}.proj_edge.weight"] = (CARD, SD)
                if h.normalize:
                    _bn_shapes(s, f"{q}.concept_reasoner{j}.bn", CARD)
                s[f"{q}.word_conv{j}.weight"] = (SD, h.nef, 1)
                for nm in ("gamma", "beta"):
                    s[f"{q}.{nm}{j}_gconv.weight"] = (gw, gc + SD, 1, 1); s[f"{q}.{nm}{j}_gconv.bias"] = (gw,)
            s[f"{p}.conv_out1.weight"] = (co, gw, 1, 1); s[f"{p}.conv_out1.bias"] = (co,)
        if ci != co:
            s[f"{p}.c_sc.weight"] = (co, ci, 1, 1); s[f"{p}.c_sc.bias"] = (co,)
    s["conv_out.1.weight"] = (3, a["out_channels"][-1], 3, 3); s["conv_out.1.bias"] = (3,)
    return s


def word_in_netg_shapes(h: Hyper):
    """state_dict of the REPAIRED concept_gan.InNetG (concept_gan.py:67-103): stem and the two ResBlockUp as OutNetG's, then
    ICAttnResBlockUp (123-165) with the word-region InConceptBlock (168-191).  Repair 1 (see `word_in_netg_forward`): the samplers'
    key projection is a grouped Conv1d over `nef` word channels per concept -- upstream builds it for noise_dim + nef (183,185,527) and
    feeds it nef (570-573), which raises; this is the only key whose shape differs from what upstream's constructor registers."""
    a = word_gen_arch(h.img_size, h.nch)
    gc = h.noise_dim + h.nef
    gw, sw = CARD * PW, CARD * SD
    full = word_netg_shapes(h)
    s = {k: v for k, v in full.items() if k.startswith(("proj_", "upblocks.0.", "upblocks.1."))}
    for i in range(2, a["depth"]):
        p, ci, co = f"upblocks.{i}", a["in_channels"][i], a["out_channels"][i]
        q = f"{p}.concept1"
        s[f"{q}.split_conv.weight"] = (gw, ci, 1, 1)
        s[f"{q}.trans_gconv.weight"] = (gw, PW, 3, 3)
        if h.normalize:
            s[f"{q}.gn.weight"] = (gw,); s[f"{q}.gn.bias"] = (gw,)
        for j in (1, 2):
            s[f"{q}.concept_sampler{j}.query_gconv.weight"] = (sw, PW, 1, 1)
            s[f"{q}.concept_sampler{j}.key_gconv.weight"] = (sw, h.nef, 1)                 # repair 1: nef, not noise_dim + nef
            if h.normalize:
                for g_ in ("gn1", "gn2"):
                    s[f"{q}.concept_sampler{j}.{g_}.weight"] = (sw,); s[f"{q}.concept_sampler{j}.{g_}.bias"] = (sw,)
            s[f"{q}.concept_reasoner{j}.proj_edge.weight"] = (CARD, SD)
            if h.normalize:
                _bn_shapes(s, f"{q}.concept_reasoner{j}.bn", CARD)
        for j in (1, 2):            # (registration order upstream: both samplers / reasoners first, then the four heads, 188-191)
            for nm in ("gamma", "beta"):
                s[f"{q}.{nm}{j}_gconv.weight"] = (gw, gc + SD, 1, 1); s[f"{q}.{nm}{j}_gconv.bias"] = (gw,)
        s[f"{p}.conv_out1.weight"] = (co, gw, 1, 1); s[f"{p}.conv_out1.bias"] = (co,)
        if ci != co:
            s[f"{p}.c_sc.weight"] = (co, ci, 1, 1); s[f"{p}.c_sc.bias"] = (co,)
    s["conv_out.1.weight"] = full["conv_out.1.weight"]; s["conv_out.1.bias"] = full["conv_out.1.bias"]
    return s


def gen_shapes(h: Hyper):
    if h.gen == "CONCEPT_OUTATTN_GEN":
        return word_netg_shapes(h)
    if h.gen == "CONCEPT_INATTN_GEN":
        return word_in_netg_shapes(h)
    return netg_shapes(h) if h.gen == "DF_GEN" else concept_netg_shapes(h)


# ----------------------------------------------------------------------------------------
# deterministic, order-independent parameter synthesis (shared by goldens and GPU tests)
# ----------------------------------------------------------------------------------------
def synth_params(shapes: dict, seed: int = 0) -> dict:
    """Kaiming-scaled weights (cf. weight_init, train_gan.py:65-69) but with non-zero biases
    and block gammas so every branch contributes; each tensor is drawn from its own
    generator seeded by crc32(key), so the values do not depend on construction order."""
    out = {}
    for key, shape in shapes.items():
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)
        leaf = key.rsplit(".", 1)[-1]
        if key.endswith(".norm"):
            t = torch.rsqrt(torch.tensor(float(SD)))
        elif leaf == "num_batches_tracked":
            t = torch.zeros((), dtype=torch.int64)
        elif leaf == "running_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            t = 0.1 * torch.randn(shape, generator=g)
        elif ".bn" in key and leaf == "weight":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif leaf == "gamma":
            t = 0.25 + 0.5 * torch.rand(shape, generator=g)
        elif ".gn" in key and leaf == "weight":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif ".fc_gamma.linear2.bias" in key:
            t = 1.0 + 0.05 * torch.randn(shape, generator=g)      # modulation scale ~ 1 (+ a data-dependent part)
        elif leaf == "bias":
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
            if ".linear2.weight" in key and ".affine" in key:
                t = t * 0.15                                       # keep activations O(1) through 5-7 blocks
            if key == "conv_out.1.weight":
                t = t * 0.35                                       # tanh mostly unsaturated
        out[key] = t if leaf == "num_batches_tracked" else t.to(torch.float32)
    return out


def ref_init_params(shapes: dict, seed: int = 0, gamma: float = 0.0) -> dict:
    """The reference's own parameterisation at the start of training: `weight_init` (train_gan.py:65-69) draws every Conv2d /
    Linear weight Kaiming-normal (fan_in, relu gain) and zeroes every bias -- the conditioning MLPs' identity init
    (df_gan.py:244-248) included, it runs after the constructors (477-478) -- and the block gammas start at `gamma` = 0
    (df_gan.py:195,281).  bench.py times this parameterisation with gamma = 0.1 so that no branch is dead weight.
    DF_GEN / DF_DISC keys only; per-key generators as in synth_params."""
    out = {}
    for key, shape in shapes.items():
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 104729 * seed) & 0x7FFFFFFF)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "gamma":
            t = torch.full(shape, float(gamma))
        elif leaf == "bias":
            t = torch.zeros(shape)
        elif leaf == "weight":
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        else:
            raise KeyError(f"ref_init_params: no rule for {key}")
        out[key] = t.to(torch.float32)
    return out


def synth_batch(h: Hyper, batch: int, seed: int = 100, words_len: int = 20):
    """COCO-shaped synthetic batch (shapes per dataset.py:34-37, encoder.py:149)."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.rand(batch, 3, h.img_size, h.img_size, generator=g) * 2 - 1
    sent = torch.randn(batch, h.text_dim, generator=g)
    words = torch.randn(batch, h.text_dim, words_len, generator=g)
    lens = torch.randint(5, words_len + 1, (batch,), generator=g)
    mask = torch.arange(words_len)[None, :] >= lens[:, None]
    noise = torch.randn(batch, h.noise_dim, generator=g)
    return dict(imgs=imgs, sent_embs=sent, words_embs=words, mask=mask, noise=noise)


# ----------------------------------------------------------------------------------------
# DF_GEN forward (model/df_gan.py:91-103, 199-224, 250-263)
# ----------------------------------------------------------------------------------------
def _affine(P, p, x, c):
    """affine.forward (df_gan.py:250-263): two 2-layer MLPs give per-sample channel scale/shift."""
    def mlp(br):
        hdn = F.relu(F.linear(c, P[f"{p}.{br}.linear1.weight"], P[f"{p}.{br}.linear1.bias"]))
        return F.linear(hdn, P[f"{p}.{br}.linear2.weight"], P[f"{p}.{br}.linear2.bias"])
    w, b = mlp("fc_gamma"), mlp("fc_beta")
    return w[:, :, None, None] * x + b[:, :, None, None]


def _g_block(P, p, x, c, upsample):
    """G_Block.forward (df_gan.py:199-224)."""
    if _QUANT:
        return _g_block_q(P, p, x, c, upsample)
    hdn = F.leaky_relu(_affine(P, f"{p}.affine0", x, c), LRELU)
    hdn = F.leaky_relu(_affine(P, f"{p}.affine1", hdn, c), LRELU)
    hdn = F.conv2d(hdn, P[f"{p}.c1.weight"], P[f"{p}.c1.bias"], 1, 1)
    hdn = F.leaky_relu(_affine(P, f"{p}.affine2", hdn, c), LRELU)
    hdn = F.leaky_relu(_affine(P, f"{p}.affine3", hdn, c), LRELU)
    res = F.conv2d(hdn, P[f"{p}.c2.weight"], P[f"{p}.c2.bias"], 1, 1)
    sc = F.conv2d(x, P[f"{p}.c_sc.weight"], P[f"{p}.c_sc.bias"]) if f"{p}.c_sc.weight" in P else x
    out = sc + P[f"{p}.gamma"] * res
    if upsample:
        out = F.interpolate(out, scale_factor=2)   # nearest (df_gan.py:202)
    return out


def _upconv3x3_q(x_lo, w, b, L=None):
    """conv3x3(nearest_up2(x_lo), w) + b the way the engine evaluates it: per output parity (i, j) a 2x2-tap convolution on the
    LOW-resolution tensor whose weights are sums of the 3x3 taps that read the same low-resolution pixel, summed in f32 and
    rounded to bf16 once (the same function as the reference's interpolate -> conv; only the weight rounding differs)."""
    B, C, H, W = x_lo.shape
    out = x_lo.new_zeros(B, w.size(0), 2 * H, 2 * W)
    rows = {0: ((0, 0), (1, 2)), 1: ((0, 1), (2, 2))}          # parity -> (kh range of th=0, kh range of th=1)
    parts = []
    for i in (0, 1):
        for j in (0, 1):
            taps = []
            for th in (0, 1):
                for tw in (0, 1):
                    (h0, h1), (w0, w1) = rows[i][th], rows[j][tw]
                    taps.append(w[:, :, h0:h1 + 1, w0:w1 + 1].sum(dim=(2, 3)))
            wc = qw(torch.stack(taps, dim=2).view(w.size(0), C, 2, 2), "g.w", None if L is None else L + ".c1")
            xp = F.pad(x_lo, (1 - j, j, 1 - i, i))              # rows a-1..a (i=0) or a..a+1 (i=1), same for columns
            parts.append((i, j, F.conv2d(xp, wc)))
    out = torch.stack([torch.stack([parts[0][2], parts[1][2]], dim=-1), torch.stack([parts[2][2], parts[3][2]], dim=-1)], dim=-3)
    out = out.reshape(B, w.size(0), 2 * H, 2 * W)               # [B,Co,H,2(i),W,2(j)] -> [B,Co,2H,2W]
    return out + b[None, :, None, None]


def _g_block_q(P, p, x, c, upsample, x_is_lo=False):
    """G_Block with the engine's storage points: one bf16 rounding after each affine-affine-LeakyReLU pass, after each
    convolution, after the 1x1 shortcut and after the block sum; `x_is_lo`: x is the previous block's output BEFORE its nearest
    upsample (the engine never writes the upsampled tensor; affines and the 1x1 shortcut commute with it, c1 runs as the fused
    upsample convolution)."""
    L = "b" + p.split(".")[-1]                  # layer tag of the per-layer ladder ("g.w@b4.sc", "g.sum@b4")
    hdn = F.leaky_relu(_affine(P, f"{p}.affine0", x, c), LRELU)
    hdn = q(F.leaky_relu(_affine(P, f"{p}.affine1", hdn, c), LRELU), "g.aff", L)
    if x_is_lo:
        hdn = q(_upconv3x3_q(hdn, P[f"{p}.c1.weight"], P[f"{p}.c1.bias"], L), "g.c1", L)
    else:
        hdn = q(F.conv2d(hdn, qw(P[f"{p}.c1.weight"], "g.w", L + ".c1"), P[f"{p}.c1.bias"], 1, 1), "g.c1", L)
    hdn = F.leaky_relu(_affine(P, f"{p}.affine2", hdn, c), LRELU)
    hdn = q(F.leaky_relu(_affine(P, f"{p}.affine3", hdn, c), LRELU), "g.aff", L)
    res = q(F.conv2d(hdn, qw(P[f"{p}.c2.weight"], "g.w", L + ".c2"), P[f"{p}.c2.bias"], 1, 1), "g.c2", L)
    # (precise trunk: the learned shortcut's weights exact -- a hi + lo pair in the engine)
    sc = q(F.conv2d(x, P[f"{p}.c_sc.weight"] if _QPRECISE else qw(P[f"{p}.c_sc.weight"], "g.w", L + ".sc"), P[f"{p}.c_sc.bias"]), "g.sc", L) \
        if f"{p}.c_sc.weight" in P else x
    if x_is_lo:
        sc = F.interpolate(sc, scale_factor=2)
    return sc, res            # the caller forms sc + gamma * res (the last block fuses the tail's LeakyReLU into that pass)


def proj_sent(P, sent):
    """netG.proj_sent: Linear(E->NEF) or Identity (df_gan.py:74-75)."""
    if "proj_sent.weight" in P:
        return F.linear(sent, P["proj_sent.weight"], P["proj_sent.bias"])
    return sent


def _stem(P, h: Hyper, noise):
    out = q(F.linear(noise, P["proj_noise.weight"], P["proj_noise.bias"]), "g.stem")      # f32 GEMM in the engine, stored as activation
    return out.view(out.size(0), 8 * h.nch, 4, 4)


def _tail(P, x, lrelu_done=False):
    """conv_out = LeakyReLU -> Conv3x3(->3) -> Tanh (df_gan.py:84-88)."""
    if not lrelu_done:
        x = q(F.leaky_relu(x, LRELU), "g.act")
    return q(torch.tanh(F.conv2d(x, qw(P["conv_out.1.weight"], "g.w", "out"), P["conv_out.1.bias"], 1, 1)), "g.img")


def netg_forward(P, h: Hyper, noise, sent_embs, **_):
    a = gen_arch(h.img_size, h.nch)
    out = _stem(P, h, noise)
    c = proj_sent(P, sent_embs)
    if _QUANT:
        lo = False
        for i in range(a["depth"]):
            p = f"upblocks.{i}"
            sc, res = _g_block_q(P, p, out, c, a["upsample"][i], x_is_lo=lo)
            out = sc + P[f"{p}.gamma"] * res
            last = i == a["depth"] - 1 and not a["upsample"][i]
            out = q(F.leaky_relu(out, LRELU), "g.sum", f"b{i}") if last else q(out, "g.sum", f"b{i}")
            lo = a["upsample"][i]
        if lo:
            out = F.interpolate(out, scale_factor=2)
        return _tail(P, out, lrelu_done=last)
    for i in range(a["depth"]):
        out = _g_block(P, f"upblocks.{i}", out, c, a["upsample"][i])
    return _tail(P, out)


# ----------------------------------------------------------------------------------------
# attention-modulation generators (model/df_concept_gan.py)
# ----------------------------------------------------------------------------------------
def _gn(P, p, x, groups):
    return F.group_norm(x, groups, P[f"{p}.weight"], P[f"{p}.bias"])


q_ = q          # (the samplers below name their query `q`)


def _cond_sampler(P, p, x, sent, normalize):
    """CondConceptSampler.forward (df_concept_gan.py:273-302): sentence query, region keys,
    softmax over H*W per (sample, concept), attention-weighted sum of x, grouped value proj."""
    B, _, H, W = x.shape
    q = sent.view(B, 1, -1).repeat(1, CARD, 1).view(B, -1, 1, 1)
    q = F.conv2d(q, P[f"{p}.query_gconv.weight"], groups=CARD)
    if normalize:
        q = _gn(P, f"{p}.gn1", q, CARD)
    q = q.view(B, CARD, -1, 1)
    k = q_(F.conv2d(x, qw(P[f"{p}.key_gconv.weight"], "g.w"), groups=CARD), "g.c.key")
    if normalize:
        k = q_(_gn(P, f"{p}.gn2", k, CARD), "g.c.keyn")
    k = k.view(B, CARD, -1, H * W)
    attn = torch.softmax(torch.matmul(q.transpose(2, 3), k), dim=3)          # [B,C,1,HW]
    ctx = torch.matmul(attn, x.view(B, CARD, -1, H * W).transpose(2, 3))      # [B,C,1,p]
    return F.conv2d(ctx.reshape(B, -1, 1, 1), P[f"{p}.value_gconv.weight"], groups=CARD)


def _self_sampler(P, p, x, normalize):
    """ConceptSampler.forward (df_concept_gan.py:554-581): query from global average pool,
    scores scaled by rsqrt(state_dim) buffer."""
    B, _, H, W = x.shape
    q = F.conv2d(F.adaptive_avg_pool2d(x, 1), P[f"{p}.query_gconv.weight"], groups=CARD)
    if normalize:
        q = _gn(P, f"{p}.gn1", q, CARD)
    q = q.view(B, CARD, 1, -1)
    k = q_(F.conv2d(x, qw(P[f"{p}.key_gconv.weight"], "g.w"), groups=CARD), "g.c.key")
    if normalize:
        k = q_(_gn(P, f"{p}.gn2", k, CARD), "g.c.keyn")
    k = k.view(B, CARD, -1, H * W)
    attn = torch.matmul(q, k).view(B, CARD, -1) * P[f"{p}.norm"]
    attn = torch.softmax(attn, dim=2).view(B, CARD, 1, H * W)
    ctx = torch.matmul(attn, x.view(B, CARD, -1, H * W).transpose(2, 3))
    return F.conv2d(ctx.reshape(B, -1, 1, 1), P[f"{p}.value_gconv.weight"], groups=CARD)


def _reasoner(P, p, x):
    """ConceptReasoner.forward (df_concept_gan.py:313-326); its normalize flag is forced False (308)."""
    B = x.size(0)
    x = x.view(B, CARD, -1)
    adj = torch.tanh(F.linear(x, P[f"{p}.proj_edge.weight"]))
    return F.relu(x + torch.matmul(adj, x)).view(B, -1, 1, 1)


def _mod_mlp(P, p, cond):
    hdn = F.leaky_relu(F.conv2d(cond, P[f"{p}.0.weight"], P[f"{p}.0.bias"], groups=CARD), LRELU)
    return F.conv2d(hdn, P[f"{p}.2.weight"], P[f"{p}.2.bias"], groups=CARD)


def _concept_block(P, p, x, sent, h: Hyper, kind):
    """InConceptBlock.residual (df_concept_gan.py:213-253) / OutConceptBlock.residual (481-531)."""
    B = x.size(0)
    e = q(F.leaky_relu(F.conv2d(x, qw(P[f"{p}.split_conv.weight"], "g.w")), LRELU), "g.c.split")
    e = q(F.conv2d(e, qw(P[f"{p}.trans_gconv.weight"], "g.w"), None, 1, 1, 1, CARD), "g.c.trans")
    if h.normalize:
        e = _gn(P, f"{p}.gn", e, CARD)
    e = q(F.leaky_relu(e, LRELU), "g.c.trunk")
    gc = sent.view(B, 1, -1).repeat(1, CARD, 1)
    out = e
    for j in (1, 2):
        if kind == "in":
            ctx = _cond_sampler(P, f"{p}.concept_sampler{j}", out, sent, h.normalize)
            ctx = _reasoner(P, f"{p}.concept_reasoner{j}", ctx).view(B, CARD, -1)
        else:
            st = _self_sampler(P, f"{p}.concept_sampler{j}", out, h.normalize)
            st = _reasoner(P, f"{p}.concept_reasoner{j}", st).view(B, CARD, -1).transpose(1, 2)  # [B,p',C]
            s = F.linear(sent, P[f"{p}.sent_linear{j}.weight"]).view(B, -1, 1)                  # [B,p',1]
            attn = torch.softmax(torch.matmul(s.transpose(1, 2), st), dim=2)                     # [B,1,C] (475-476)
            ctx = (st * attn).transpose(1, 2)                                                    # [B,C,p']
        cond = torch.cat([gc, ctx], dim=2).reshape(B, -1, 1, 1)
        gamma = _mod_mlp(P, f"{p}.gamma{j}_gconv", cond)
        beta = _mod_mlp(P, f"{p}.beta{j}_gconv", cond)
        out = q(F.leaky_relu(gamma * out + beta, LRELU), "g.c.mod")
    return out


def _concept_g_block(P, p, x, sent, h: Hyper, kind, upsample):
    """ICAttnG_Block (df_concept_gan.py:133-156) / OCAG_Block (395-418)."""
    pad = 1 if kind == "in" else 0
    r = _concept_block(P, f"{p}.concept1", x, sent, h, kind)
    r = q(F.leaky_relu(F.conv2d(r, qw(P[f"{p}.conv_out1.weight"], "g.w"), P[f"{p}.conv_out1.bias"], 1, pad), LRELU), "g.c.out1")
    r = _concept_block(P, f"{p}.concept2", r, sent, h, kind)
    r = q(F.conv2d(r, qw(P[f"{p}.conv_out2.weight"], "g.w"), P[f"{p}.conv_out2.bias"], 1, pad), "g.c.out2")
    sc = q(F.conv2d(x, qw(P[f"{p}.c_sc.weight"], "g.w"), P[f"{p}.c_sc.bias"]), "g.sc") if f"{p}.c_sc.weight" in P else x
    out = q(P[f"{p}.gamma"] * r + sc, "g.sum")
    if upsample:
        out = F.interpolate(out, scale_factor=2)
    return out


def concept_netg_forward(P, h: Hyper, noise, sent_embs, **_):
    kind = {"CONCEPT_IN_DF_GEN": "in", "CONCEPT_OUT_DF_GEN": "out"}[h.gen]
    a = gen_arch(h.img_size, h.nch)
    c = proj_sent(P, sent_embs)
    out = _stem(P, h, noise)
    for i in range(a["depth"]):
        out = _concept_g_block(P, f"upblocks.{i}", out, c, h, kind, a["upsample"][i])
    return _tail(P, out)


# ----------------------------------------------------------------------------------------
# word-attention generator (model/concept_gan.py OutNetG; un-wired upstream: its registry names are commented out,
# train_gan.py:31,44)
# ----------------------------------------------------------------------------------------
def _bn(P, p, x, train):
    """nn.BatchNorm{1,2}d: batch statistics + running-stat update (momentum 0.1, unbiased variance) when training"""
    if train:
        P[f"{p}.num_batches_tracked"] += 1
    return F.batch_norm(x, P[f"{p}.running_mean"], P[f"{p}.running_var"], P[f"{p}.weight"], P[f"{p}.bias"], train, 0.1, 1e-5)


def _word_reasoner(P, p, x, normalize, train):
    """concept_gan.ConceptReasoner.forward (640-653): unlike df_concept_gan's, BatchNorm1d over the 16 concepts is live"""
    B = x.size(0)
    x = x.view(B, CARD, -1)
    adj = torch.tanh(F.linear(x, P[f"{p}.proj_edge.weight"]))
    out = x + torch.matmul(adj, x)
    if normalize:
        out = _bn(P, f"{p}.bn", out, train)
    return F.relu(out).view(B, -1, 1, 1)


def _word_context(state, words, mask):
    """OutConceptBlock.get_context_embs (374-394).  state [B,C,p'] is L2-normalised over dim 1 -- the CONCEPT axis, as
    written upstream -- words [B,p',T] over p'; cosine-like scores [B,C,T], padded words masked to -inf, softmax over T,
    attention-weighted sum of the normalised word vectors -> [B,C,p']."""
    st = F.normalize(state, p=2, dim=1)
    wd = F.normalize(words, p=2, dim=1)
    sim = torch.matmul(st, wd).masked_fill(mask.view(mask.size(0), 1, -1), float("-inf"))
    return torch.matmul(torch.softmax(sim, dim=2), wd.transpose(1, 2))


def _word_concept_block(P, p, x, gcond, words, mask, h: Hyper, upsample, train):
    """OutConceptBlock.forward (396-449).  Second stage as upstream has it: concept_sampler2's result is discarded, and
    concept_reasoner2 is applied to the first CONTEXT but its result is discarded too (431-433) -- its only effect is the
    BatchNorm running-statistics update; the second word attention queries with the first context."""
    B = x.size(0)
    e = F.relu(F.conv2d(x, P[f"{p}.split_conv.weight"]))
    e = F.conv2d(e, P[f"{p}.trans_gconv.weight"], None, 1, 1, 1, CARD)
    if h.normalize:
        e = _gn(P, f"{p}.gn", e, CARD)
    e = F.relu(e)
    st = _self_sampler(P, f"{p}.concept_sampler1", e, h.normalize)
    st = _word_reasoner(P, f"{p}.concept_reasoner1", st, h.normalize, train).view(B, CARD, -1)
    ctx = _word_context(st, F.conv1d(words, P[f"{p}.word_conv1.weight"]), mask)                       # [B,C,p']
    gc = gcond.view(B, 1, -1).repeat(1, CARD, 1)
    cond = torch.cat([gc, ctx], dim=2).reshape(B, -1, 1, 1)
    gamma = F.conv2d(cond, P[f"{p}.gamma1_gconv.weight"], P[f"{p}.gamma1_gconv.bias"], groups=CARD)
    beta = F.conv2d(cond, P[f"{p}.beta1_gconv.weight"], P[f"{p}.beta1_gconv.bias"], groups=CARD)
    out = F.relu(gamma * e + beta)
    if upsample:
        out = F.interpolate(out, scale_factor=2)
    if h.normalize and train:
        with torch.no_grad():
            _word_reasoner(P, f"{p}.concept_reasoner2", ctx.reshape(B, -1, 1, 1), True, True)
    ctx2 = _word_context(ctx, F.conv1d(words, P[f"{p}.word_conv2.weight"]), mask)
    cond = torch.cat([gc, ctx2], dim=2).reshape(B, -1, 1, 1)
    gamma = F.conv2d(cond, P[f"{p}.gamma2_gconv.weight"], P[f"{p}.gamma2_gconv.bias"], groups=CARD)
    beta = F.conv2d(cond, P[f"{p}.beta2_gconv.weight"], P[f"{p}.beta2_gconv.bias"], groups=CARD)
    return F.relu(gamma * out + beta)


def _res_block_up(P, p, x, gcond, h: Hyper, train):
    """ResBlockUp (454-512) as OutNetG builds its first two blocks: `upsample` receives the whole arch list (262), which
    is truthy, so both upsample."""
    g1 = F.linear(gcond, P[f"{p}.linear_gamma1.weight"])[:, :, None, None]
    b1 = F.linear(gcond, P[f"{p}.linear_beta1.weight"])[:, :, None, None]
    o = _bn(P, f"{p}.bn1", x, train) if h.normalize else x
    o = F.interpolate(F.relu(g1 * o + b1), scale_factor=2)
    o = F.conv2d(o, P[f"{p}.c1.weight"], P[f"{p}.c1.bias"], 1, 1)
    g2 = F.linear(gcond, P[f"{p}.linear_gamma2.weight"])[:, :, None, None]
    b2 = F.linear(gcond, P[f"{p}.linaer_beta2.weight"])[:, :, None, None]
    if h.normalize:
        o = _bn(P, f"{p}.bn2", o, train)
    o = F.conv2d(F.relu(g2 * o + b2), P[f"{p}.c2.weight"], P[f"{p}.c2.bias"], 1, 1)
    sc = F.interpolate(x, scale_factor=2)
    if f"{p}.c_sc.weight" in P:
        sc = F.conv2d(sc, P[f"{p}.c_sc.weight"], P[f"{p}.c_sc.bias"])
    return o + sc


def word_netg_forward(P, h: Hyper, noise, sent_embs, words_embs=None, mask=None, train=True, **_):
    """concept_gan.OutNetG.forward (281-298)"""
    a = word_gen_arch(h.img_size, h.nch)
    sent = F.linear(sent_embs, P["proj_sent.weight"], P["proj_sent.bias"])
    words = F.conv1d(words_embs, P["proj_word.weight"], P["proj_word.bias"])
    gcond = torch.cat([noise, sent], dim=1)
    out = F.linear(gcond, P["proj_cond.weight"], P["proj_cond.bias"]).view(noise.size(0), -1, 4, 4)
    for i in range(a["depth"]):
        p = f"upblocks.{i}"
        if i < 2:
            out = _res_block_up(P, p, out, gcond, h, train)
            continue
        up = a["upsample"][i]
        r = _word_concept_block(P, f"{p}.concept1", out, gcond, words, mask, h, up, train)
        r = F.conv2d(r, P[f"{p}.conv_out1.weight"], P[f"{p}.conv_out1.bias"])
        sc = F.interpolate(out, scale_factor=2) if up else out
        if f"{p}.c_sc.weight" in P:
            sc = F.conv2d(sc, P[f"{p}.c_sc.weight"], P[f"{p}.c_sc.bias"])
        out = r + sc
    return _tail(P, out)


def _word_region_sampler(P, p, x, words, mask, normalize):
    """concept_gan.CondConceptSampler.forward + get_context_embs (532-580): every REGION queries the caption words.  query =
    grouped 1x1 of the map [-> GroupNorm], key = grouped Conv1d of the words repeated per concept [-> GroupNorm over (p', T), padded
    positions included, as upstream]; both L2-normalised over p'; cosine scores [B,C,HW,T], padded words -> -inf, softmax over T,
    attention-weighted sum of the normalised keys, MEAN over the regions -> [B, C*p', 1, 1]."""
    B, _, H, W = x.shape
    T = words.size(-1)
    qy = F.conv2d(x, P[f"{p}.query_gconv.weight"], groups=CARD)
    if normalize:
        qy = _gn(P, f"{p}.gn1", qy, CARD)
    qy = F.normalize(qy.view(B, CARD, -1, H * W), p=2, dim=2)
    k = F.conv1d(words.view(B, 1, -1, T).repeat(1, CARD, 1, 1).view(B, -1, T), P[f"{p}.key_gconv.weight"], groups=CARD)
    if normalize:
        k = _gn(P, f"{p}.gn2", k, CARD)
    k = F.normalize(k.view(B, CARD, -1, T), p=2, dim=2)
    sim = torch.matmul(qy.transpose(2, 3), k).masked_fill(mask.view(B, 1, 1, T), float("-inf"))
    ctx = torch.matmul(torch.softmax(sim, dim=3), k.transpose(2, 3)).mean(dim=2)                   # [B,C,p']
    return ctx.reshape(B, -1, 1, 1)


def _word_in_concept_block(P, p, x, gcond, words, mask, h: Hyper, upsample, train):
    """concept_gan.InConceptBlock.forward (193-240) with repair 2: `self.upsample`, read at 222 and never assigned by the constructor
    (170-191), is the enclosing ICAttnResBlockUp's flag -- the value its shortcut (148-149) needs the residual to agree with."""
    B = x.size(0)
    e = F.relu(F.conv2d(x, P[f"{p}.split_conv.weight"]))
    e = F.conv2d(e, P[f"{p}.trans_gconv.weight"], None, 1, 1, 1, CARD)
    if h.normalize:
        e = _gn(P, f"{p}.gn", e, CARD)
    out = F.relu(e)
    gc = gcond.view(B, 1, -1).repeat(1, CARD, 1)
    for j in (1, 2):
        ctx = _word_region_sampler(P, f"{p}.concept_sampler{j}", out, words, mask, h.normalize)
        ctx = _word_reasoner(P, f"{p}.concept_reasoner{j}", ctx, h.normalize, train).view(B, CARD, -1)
        cond = torch.cat([gc, ctx], dim=2).reshape(B, -1, 1, 1)
        gamma = F.conv2d(cond, P[f"{p}.gamma{j}_gconv.weight"], P[f"{p}.gamma{j}_gconv.bias"], groups=CARD)
        beta = F.conv2d(cond, P[f"{p}.beta{j}_gconv.weight"], P[f"{p}.beta{j}_gconv.bias"], groups=CARD)
        out = F.relu(gamma * out + beta)
        if j == 1 and upsample:
            out = F.interpolate(out, scale_factor=2)
    return out


def word_in_netg_forward(P, h: Hyper, noise, sent_embs, words_embs=None, mask=None, train=True, **_):
    """concept_gan.InNetG.forward (105-121), REPAIRED.  Upstream's class cannot run: (1) CondConceptSampler.key_gconv is
    built with cond_dim = noise_dim + nef input channels per concept (137, 183, 527) but receives the projected words, nef channels
    (570-573) -> RuntimeError; (2) InConceptBlock.forward reads self.upsample (222), which no constructor sets -> AttributeError.
    The repair changes exactly those two things -- key_gconv takes nef channels (what it is fed), the block inherits its parent's
    upsample flag (what the parent's shortcut assumes) -- and nothing else; `oracle/make_golden.py` applies the same two patches to
    the reference's own objects at run time, so fwd_wordin* / step_wordin* pin this restatement against upstream's code for every
    other line."""
    a = word_gen_arch(h.img_size, h.nch)
    sent = F.linear(sent_embs, P["proj_sent.weight"], P["proj_sent.bias"])
    words = F.conv1d(words_embs, P["proj_word.weight"], P["proj_word.bias"])
    gcond = torch.cat([noise, sent], dim=1)
    out = F.linear(gcond, P["proj_cond.weight"], P["proj_cond.bias"]).view(noise.size(0), -1, 4, 4)
    for i in range(a["depth"]):
        p = f"upblocks.{i}"
        if i < 2:
            out = _res_block_up(P, p, out, gcond, h, train)
            continue
        up = a["upsample"][i]
        r = _word_in_concept_block(P, f"{p}.concept1", out, gcond, words, mask, h, up, train)
        r = F.conv2d(r, P[f"{p}.conv_out1.weight"], P[f"{p}.conv_out1.bias"])
        sc = F.interpolate(out, scale_factor=2) if up else out
        if f"{p}.c_sc.weight" in P:
            sc = F.conv2d(sc, P[f"{p}.c_sc.weight"], P[f"{p}.c_sc.bias"])
        out = r + sc
    return _tail(P, out)


def gen_forward(P, h: Hyper, noise, sent_embs, **kw):
    if h.gen == "CONCEPT_OUTATTN_GEN":
        return word_netg_forward(P, h, noise, sent_embs, **kw)
    if h.gen == "CONCEPT_INATTN_GEN":
        return word_in_netg_forward(P, h, noise, sent_embs, **kw)
    f = netg_forward if h.gen == "DF_GEN" else concept_netg_forward
    return f(P, h, noise, sent_embs, **kw)


# ----------------------------------------------------------------------------------------
# DF_DISC forward (model/df_gan.py:125-132, 283-294) and COND_DNET (162-176)
# ----------------------------------------------------------------------------------------
def sn_weight(P, name, train=True, eps=1e-12):
    """Effective weight of a layer: P[name], or -- when the layer is spectrally normalised (P has name + "_orig") -- what the
    forward pre-hook of torch.nn.utils.spectral_norm computes (the legacy hook implementation the reference imports,
    modules.py:3,16-17): in training mode ONE power iteration per forward call, v <- normalize(W^T u), u <- normalize(W v),
    written back into the buffers; sigma = u . (W v) with u, v treated as constants; W / sigma."""
    if name + "_orig" not in P:
        return P[name]
    W, u, v = P[name + "_orig"], P[name + "_u"], P[name + "_v"]
    Wm = W.reshape(W.size(0), -1)
    if train:
        with torch.no_grad():
            v_new = F.normalize(torch.mv(Wm.t(), u), dim=0, eps=eps)
            u_new = F.normalize(torch.mv(Wm, v_new), dim=0, eps=eps)
            v.copy_(v_new)
            u.copy_(u_new)
        u, v = u.clone(), v.clone()
    sigma = torch.dot(u, torch.mv(Wm, v))
    return W / sigma


def netd_forward(P, h: Hyper, x, second_order=False):
    a = disc_arch(h.img_size, h.nch)
    if _QUANT:
        return _netd_forward_q(P, h, x, a, second_order)
    out = F.conv2d(x, sn_weight(P, "conv_img.weight"), P["conv_img.bias"], 1, 1)
    for i in range(1, a["depth"]):
        p = f"downblocks.{i - 1}"
        r = F.leaky_relu(F.conv2d(out, sn_weight(P, f"{p}.conv_r.0.weight"), None, 2, 1), LRELU)
        r = F.leaky_relu(F.conv2d(r, sn_weight(P, f"{p}.conv_r.2.weight"), None, 1, 1), LRELU)
        s = out
        if a["cin"][i] != a["cout"][i]:                     # learned_shortcut (df_gan.py:270,287)
            s = F.conv2d(s, sn_weight(P, f"{p}.conv_s.weight"), P[f"{p}.conv_s.bias"])
        s = F.avg_pool2d(s, 2)
        out = s + P[f"{p}.gamma"] * r
    return out


def _dstem_compose_q(P):
    """The engine's composed discriminator stem (round 4, csrc/dstem.hip): conv_img feeds the first block without an activation, so
    conv_r[0] o conv_img is a 6x6 stride-2 pad-2 convolution of the image and conv_s o avg_pool2d o conv_img a 4x4 stride-2 pad-1
    one (df_gan.py:114,127,272-291).  Returns (W_A [2ndf,3,6,6], b_A, W_B [2ndf,3,4,4], b_B) as differentiable functions of the five
    parameters: what the engine rounds to 16 bits ONCE, in place of conv_img's output and the two weight tensors separately."""
    wi, bi = P["conv_img.weight"], P["conv_img.bias"]
    w0, ws, bs = P["downblocks.0.conv_r.0.weight"], P["downblocks.0.conv_s.weight"], P["downblocks.0.conv_s.bias"]
    wa = F.conv_transpose2d(w0, wi)                                       # full correlation over conv_img's channels
    ba = torch.einsum("omhw,m->o", w0, bi)
    box = torch.full((1, 1, 2, 2), 0.25, dtype=wi.dtype)
    wp = F.conv_transpose2d(wi.reshape(-1, 1, 3, 3), box).reshape(wi.shape[0], wi.shape[1], 4, 4)
    wb = torch.einsum("om,mcab->ocab", ws[:, :, 0, 0], wp)
    return wa, ba, wb, ws[:, :, 0, 0] @ bi + bs


def dstem_applies(h: Hyper, x, second_order=False):
    """where the engine takes the composed stem: 16-bit modes at the width it is built for (ndf = 32), images that tile, no
    spectral norm (ops.dstem_eligible / NetD.forward); since round 4 also in the second-order pass (ops.DStemBwdFn)"""
    return h.nch == 32 and not h.spec_norm and x.shape[2] % 16 == 0 and x.shape[3] % 64 == 0


def _netd_forward_q(P, h: Hyper, x, a, second_order=False):
    """DF_DISC with the engine's storage points (image, every convolution output, the pooled shortcut input, the block sum) and
    its order on the shortcut: average pool first, then the 1x1 convolution (they commute; df_gan.py:286-291).
    ``second_order``: the pass whose backward is differentiated again (MA-GP) -- there the engine stores the residual branch and
    rounds gamma * s * dout in one pointwise pass; everywhere else it stores the branch's sign bits (_QBranchTimesGamma).
    Where the engine runs conv_img and the first block on the composed stem (`dstem_applies`) the first block's first activation
    and its shortcut come straight from the rounded image through the rounded COMPOSED weights: conv_img's output, its pooled copy
    and the separate roundings of the three weight tensors do not exist (the block's border pixels, which the engine computes with
    separately rounded correction tables, are taken from the same composed convolution of the un-composed f32 form here: 3 % of the
    map, a rounding-level difference)."""
    stem = dstem_applies(h, x, second_order) and "d.conv_img" not in _QSKIP
    xq = q(x, "d.img")
    if stem:
        wa, ba, wb, bb = _dstem_compose_q(P)
        ci = F.conv2d(xq, P["conv_img.weight"], P["conv_img.bias"], 1, 1)                       # f32, never stored: border pixels only
        r_exact = F.conv2d(ci, P["downblocks.0.conv_r.0.weight"], None, 2, 1)
        r_comp = F.conv2d(xq, qw(wa, "d.w", "b0.r0"), ba, 2, 2)
        m = torch.zeros_like(r_comp[:1, :1])
        m[:, :, 1:-1, 1:-1] = 1.0
        r0 = q(F.leaky_relu(m * r_comp + (1.0 - m) * r_exact, LRELU), "d.r0", "b0")
        s0 = q(F.conv2d(xq, qw(wb, "d.w", "b0.s"), bb, 2, 1), "d.sc", "b0")
        out = None
    else:
        out = q(F.conv2d(xq, qw(sn_weight(P, "conv_img.weight"), "d.w", "img"), P["conv_img.bias"], 1, 1), "d.conv_img")
    precise = _QPRECISE and not second_order
    trunk = None                       # precise trunk: the previous block's f32 sum (its pooled by-product is not rounded)
    for i in range(1, a["depth"]):
        p, L = f"downblocks.{i - 1}", f"b{i - 1}"              # L: the layer tag of the per-layer ladder ("d.w@b3.s", "d.sum@b3")
        last = i == a["depth"] - 1
        if stem and i == 1:
            r, s = r0, s0
            small = False
        else:
            small = precise and out.shape[2] // 2 <= 8         # the block's maps: <= 8x8 pixels
            r = q(F.leaky_relu(F.conv2d(out, qw(sn_weight(P, f"{p}.conv_r.0.weight"), "d.w", L + ".r0"), None, 2, 1), LRELU), "d.r0", L)
            s = q(F.avg_pool2d(out, 2), "d.pool", L) if trunk is None else q(F.avg_pool2d(trunk, 2), "d.pool", L, grad_only=True)
            if a["cin"][i] != a["cout"][i]:
                ws = sn_weight(P, f"{p}.conv_s.weight")
                s = q(F.conv2d(s, ws if precise else qw(ws, "d.w", L + ".s"), P[f"{p}.conv_s.bias"]), "d.sc", L, grad_only=small)
        z = F.conv2d(r, qw(sn_weight(P, f"{p}.conv_r.2.weight"), "d.w", L + ".r2"), None, 1, 1)
        site = "d.last" if (last and "d.last" in _QSKIP) else "d.sum"
        if second_order or not _rounded("d.r2", L):
            out = q(s + P[f"{p}.gamma"] * q(F.leaky_relu(z, LRELU), "d.r2", L), site, L)
            trunk = None
        elif small:
            trunk = q(s + _QBranchTimesGamma.apply(z, P[f"{p}.gamma"], False), site, L, grad_only=True)
            out = trunk if last else _bf16(trunk.detach()) + (trunk - trunk.detach())      # conv_r[0] of the next block reads the rounded sum
        else:
            out = q(s + _QBranchTimesGamma.apply(z, P[f"{p}.gamma"]), site, L)
            trunk = None
    return out


def cond_dnet(P, h: Hyper, feat, sent_embs, second_order=False):
    """D_GET_LOGITS.forward (df_gan.py:162-176) -> [logit[B,1,1,1], img_emb, txt_emb]."""
    B = feat.size(0)
    out = F.avg_pool2d(feat, 4).view(B, -1)
    has_proj = "COND_DNET.proj_match.weight" in P or "COND_DNET.proj_match.weight_orig" in P
    if h.img_match:
        out = F.linear(out, sn_weight(P, "COND_DNET.proj_match.weight"), P["COND_DNET.proj_match.bias"])
    elif has_proj:
        sent_embs = F.linear(sent_embs, sn_weight(P, "COND_DNET.proj_match.weight"), P["COND_DNET.proj_match.bias"])
    if _QPRECISE and not second_order:      # precise trunk: the last block handed over an f32 map and COND_DNET runs in f32 (not in the MA-GP pass)
        c = sent_embs.view(B, -1, 1, 1).repeat(1, 1, 4, 4)
        m = F.leaky_relu(F.conv2d(torch.cat((feat, c), 1), sn_weight(P, "COND_DNET.joint_conv.0.weight"), None, 1, 1), LRELU)
        return [F.conv2d(m, sn_weight(P, "COND_DNET.joint_conv.2.weight")), out, sent_embs]
    c = q(sent_embs, "h.c").view(B, -1, 1, 1).repeat(1, 1, 4, 4)        # engine: the condition joins the bf16 feature map
    hc = torch.cat((feat, c), 1)
    m = q(F.leaky_relu(F.conv2d(hc, qw(sn_weight(P, "COND_DNET.joint_conv.0.weight"), "h.w", "j0"), None, 1, 1), LRELU), "h.m")
    m = F.conv2d(m, qw(sn_weight(P, "COND_DNET.joint_conv.2.weight"), "h.w", "j2"))         # the logit itself leaves the engine in f32
    return [m, out, sent_embs]


# ----------------------------------------------------------------------------------------
# contrastive head (train_gan.py:72-139)
# ----------------------------------------------------------------------------------------
# ------------------------------------------------------------------ frozen text front end (encoder.py:73-153)
def rnn_encoder_shapes(voca_size: int, emb_dim: int = 256, ninput: int = 300, rnn_type: str = "LSTM") -> dict:
    """state_dict of RNN_ENCODER: nn.Embedding(V, 300) + one-layer bidirectional nn.LSTM(300, emb_dim/2) or nn.GRU
    (encoder.py:75-104).  Gate rows are ordered i, f, g, o (torch.nn.LSTM) / r, z, n (torch.nn.GRU)."""
    H = emb_dim // 2
    G = {"LSTM": 4, "GRU": 3}[rnn_type]
    out = {"encoder.weight": (voca_size, ninput)}
    for sfx in ("", "_reverse"):
        out[f"rnn.weight_ih_l0{sfx}"] = (G * H, ninput)
        out[f"rnn.weight_hh_l0{sfx}"] = (G * H, H)
        out[f"rnn.bias_ih_l0{sfx}"] = (G * H,)
        out[f"rnn.bias_hh_l0{sfx}"] = (G * H,)
    return out


def synth_rnn_params(shapes: dict, seed: int = 0) -> dict:
    """embedding uniform(-0.1, 0.1) as _init_weights (encoder.py:106-108); LSTM tensors uniform(-1/sqrt(H), 1/sqrt(H))
    as nn.LSTM.reset_parameters; per-key generators as in synth_params."""
    H = shapes["rnn.weight_hh_l0"][1]
    out = {}
    for key, shape in shapes.items():
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)
        a = 0.1 if key == "encoder.weight" else 1.0 / math.sqrt(H)
        out[key] = ((torch.rand(shape, generator=g) * 2 - 1) * a).to(torch.float32)
    return out


def synth_captions(batch: int, max_len: int, voca_size: int, seed: int = 0):
    """token ids in [1, V) followed by zero padding, lengths in [1, max_len] (dataset.py:104-111: get_caption pads
    with 0 and truncates to MAX_LENGTH); sample 0 has full length and sample 1 length 1 when the batch allows."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, max_len + 1, (batch,), generator=g)
    lens[0] = max_len
    if batch > 1:
        lens[1] = 1
    caps = torch.randint(1, voca_size, (batch, max_len), generator=g)
    caps = caps * (torch.arange(max_len)[None, :] < lens[:, None])
    return caps.to(torch.int64), lens.to(torch.int64)


def rnn_encoder(P, caps, cap_lens, n_steps: int):
    """RNN_ENCODER.forward in eval mode (encoder.py:118-153; dropout is the identity, train_gan.py:468); the cell type
    (encoder.py:95-102) follows from the parameter shapes: 4H gate rows = nn.LSTM, 3H = nn.GRU.

    The reference sorts by length, packs, runs nn.LSTM and un-sorts; per sample that is: the forward direction runs
    t = 0..len-1, the reverse direction t = len-1..0, both from zero state; outputs at t >= len are zero
    (pad_packed_sequence, total_length = n_steps); the sentence embedding is [h_fwd(len-1), h_rev(0)].
    Returns words_embs [B, 2H, n_steps], sent_embs [B, 2H], mask [B, T] (True where the token id is 0)."""
    B = caps.shape[0]
    H = P["rnn.weight_hh_l0"].shape[1]
    emb = P["encoder.weight"][caps]                                     # [B, T, 300]
    words = torch.zeros(B, 2 * H, n_steps, dtype=emb.dtype)
    sent = torch.zeros(B, 2 * H, dtype=emb.dtype)
    if P["rnn.weight_hh_l0"].shape[0] == 3 * H:
        # nn.GRU (sentence embedding = the final hidden state, encoder.py:146-147):
        #   r = s(W_ir x + b_ir + W_hr h + b_hr)   z = s(W_iz x + b_iz + W_hz h + b_hz)
        #   n = tanh(W_in x + b_in + r * (W_hn h + b_hn))   h' = (1 - z) * n + z * h
        for d, sfx in enumerate(("", "_reverse")):
            w_ih, w_hh = P[f"rnn.weight_ih_l0{sfx}"], P[f"rnn.weight_hh_l0{sfx}"]
            b_ih, b_hh = P[f"rnn.bias_ih_l0{sfx}"], P[f"rnn.bias_hh_l0{sfx}"]
            for b in range(B):
                n = int(cap_lens[b])
                hcur = torch.zeros(H, dtype=emb.dtype)
                for t in (range(n) if d == 0 else range(n - 1, -1, -1)):
                    gi, gh = w_ih @ emb[b, t] + b_ih, w_hh @ hcur + b_hh
                    r = torch.sigmoid(gi[:H] + gh[:H])
                    z = torch.sigmoid(gi[H:2 * H] + gh[H:2 * H])
                    cand = torch.tanh(gi[2 * H:] + r * gh[2 * H:])
                    hcur = (1 - z) * cand + z * hcur
                    words[b, d * H:(d + 1) * H, t] = hcur
                sent[b, d * H:(d + 1) * H] = hcur
        return words, sent, caps == 0
    for d, sfx in enumerate(("", "_reverse")):
        w_ih, w_hh = P[f"rnn.weight_ih_l0{sfx}"], P[f"rnn.weight_hh_l0{sfx}"]
        bias = P[f"rnn.bias_ih_l0{sfx}"] + P[f"rnn.bias_hh_l0{sfx}"]
        for b in range(B):
            n = int(cap_lens[b])
            hcur, ccur = torch.zeros(H, dtype=emb.dtype), torch.zeros(H, dtype=emb.dtype)
            for t in (range(n) if d == 0 else range(n - 1, -1, -1)):
                gates = w_ih @ emb[b, t] + w_hh @ hcur + bias
                i, f = torch.sigmoid(gates[:H]), torch.sigmoid(gates[H:2 * H])
                gg, o = torch.tanh(gates[2 * H:3 * H]), torch.sigmoid(gates[3 * H:])
                ccur = f * ccur + i * gg
                hcur = o * torch.tanh(ccur)
                words[b, d * H:(d + 1) * H, t] = hcur
            sent[b, d * H:(d + 1) * H] = hcur
    return words, sent, caps == 0


def cosine_scores(a, b):
    """train_gan.py:85-91."""
    return F.normalize(a, p=2, dim=1) @ F.normalize(b, p=2, dim=1).t()


def make_labels(batch_size, sent_embs, b_global, smooth_global=0.0, p=0.6):
    """train_gan.py:72-83 (note the [B]-against-[B,B] broadcast of 1/num_pos runs along columns)."""
    labels = torch.eye(batch_size)
    if b_global:
        sim = cosine_scores(sent_embs, sent_embs)
        sim.fill_diagonal_(3)
        pos = (sim > p) & (sim < 3)
        num_pos = pos.sum(1).clamp(min=1) + 1
        w = smooth_global if smooth_global != 0.0 else torch.reciprocal(num_pos.float())
        labels = (labels + w * pos).clamp(max=1)
    return labels.detach()


def contrastive_loss(a, b, labels, b_global, smooth_global=0.0):
    """sent_loss / img_loss (train_gan.py:93-115 / 117-139; identical bodies): symmetric InfoNCE,
    no temperature, column-direction then row-direction."""
    if not b_global:
        num_pos = 1
    elif smooth_global == 0.0:
        num_pos = 2
    else:
        num_pos = (labels > 0).sum(1)
    s = cosine_scores(a, b)
    l0 = (-(F.log_softmax(s, dim=0) * labels).sum(0) / num_pos).mean()
    l1 = (-(F.log_softmax(s, dim=1) * labels).sum(1) / num_pos).mean()
    return l0 + l1


# ----------------------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults: eps 1e-8, no weight decay, no amsgrad; train_gan.py:483-484)
# ----------------------------------------------------------------------------------------
@dataclass
class AdamState:
    lr: float
    betas: tuple
    eps: float = 1e-8
    step: dict = field(default_factory=dict)
    m: dict = field(default_factory=dict)
    v: dict = field(default_factory=dict)

    def apply(self, P: dict, grads: dict):
        """Update P in place for every key with a (non-None) grad, like optimizer.step()."""
        b1, b2 = self.betas
        for k, g in grads.items():
            if g is None:
                continue
            if k not in self.m:
                self.m[k] = torch.zeros_like(P[k]); self.v[k] = torch.zeros_like(P[k]); self.step[k] = 0
            self.step[k] += 1
            t = self.step[k]
            self.m[k].mul_(b1).add_(g, alpha=1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            with torch.no_grad():
                P[k].addcdiv_(self.m[k], denom, value=-(self.lr / bc1))


# ----------------------------------------------------------------------------------------
# one training iteration (train_gan.py:174-293)
# ----------------------------------------------------------------------------------------
def _leaves(P):
    """Differentiable leaf copies of the parameters; the spectral-norm power-iteration buffers are SHARED (not copied), so the
    in-place updates every forward call makes to them persist like the reference's module buffers do."""
    shared = (".weight_u", ".weight_v", ".running_mean", ".running_var", ".num_batches_tracked")   # module buffers updated in place
    return {k: (v if k.endswith(shared) else v.detach().clone().requires_grad_(not k.endswith(".norm")))
            for k, v in P.items()}


def _grads(loss, P, retain=False):
    keys = [k for k, v in P.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [P[k] for k in keys], allow_unused=True, retain_graph=retain)
    return dict(zip(keys, gs))


def train_step(PG, PD, optG: AdamState, optD: AdamState, h: Hyper, batch, it_count=1):
    """One iteration of the loop body.  PG / PD are updated in place (plain tensors).
    Returns a dict with every scalar the reference computes and the grads of each backward."""
    imgs, sent, words, mask, noise = (batch[k] for k in ("imgs", "sent_embs", "words_embs", "mask", "noise"))
    B = imgs.size(0)
    G, D = _leaves(PG), _leaves(PD)
    out = {}

    # ---- D step (187-229)
    psent = sent if h.seperate else proj_sent(G, sent)
    real_feat = netd_forward(D, h, imgs)
    o_real = cond_dnet(D, h, real_feat, psent.detach())
    errD_real = F.relu(1.0 - o_real[0]).mean()
    fake = gen_forward(G, h, noise, sent, words_embs=words, mask=mask)
    o_fake = cond_dnet(D, h, netd_forward(D, h, fake.detach()), psent.detach())
    errD_fake = F.relu(1.0 + o_fake[0]).mean()
    mis = errD_fake
    if h.rmis:
        o_mis = cond_dnet(D, h, real_feat[: B - 1], psent[1:B].detach())
        errD_mis = F.relu(1.0 + o_mis[0]).mean()
        mis = mis + errD_mis
        out["errD_mismatch"] = errD_mis.item()
    any_enc = h.enc_sent or h.enc_disc
    labels = make_labels(B, sent, h.b_global, h.smooth_global) if any_enc else None
    enc = 0.0
    if h.enc_sent:
        assert h.sent_match or h.img_match
        ds = contrastive_loss(o_real[1], o_real[2], labels, h.b_global, h.smooth_global)
        enc = enc + h.smooth_sent * ds
        out["ds_loss"] = ds.item()
    errD = errD_real + mis * h.smooth_mismatch + enc
    gD = _grads(errD, D)
    optD.apply(PD, gD)
    out.update(errD_real=errD_real.item(), errD_fake=errD_fake.item(), errD=errD.item(), grads_D=gD,
               fake=fake.detach(), logit_real=o_real[0].detach(), logit_fake=o_fake[0].detach())

    # ---- MA-GP (231-252): uses the post-step D weights, own optimizer step
    if h.magp:
        D = _leaves(PD)
        xi = imgs.detach().clone().requires_grad_()
        si = psent.detach().clone().requires_grad_()
        o = cond_dnet(D, h, netd_forward(D, h, xi, second_order=True), si, second_order=True)
        g0, g1 = torch.autograd.grad(o[0], (xi, si), torch.ones_like(o[0]), create_graph=True)
        gcat = torch.cat((g0.reshape(B, -1), g1.reshape(B, -1)), dim=1)
        gp = (gcat.pow(2).sum(1).sqrt() ** 6).mean()
        d_loss = 2.0 * gp
        gGP = _grads(d_loss, D)
        optD.apply(PD, gGP)
        out.update(d_loss_gp=gp.item(), grads_GP=gGP)

    # ---- G step (256-291)
    if it_count % h.n_critic == 0:
        D = _leaves(PD)
        feat = netd_forward(D, h, fake)
        o = cond_dnet(D, h, feat, psent)
        errG_fake = -o[0].mean()
        enc = 0.0
        if h.enc_sent:
            gs = contrastive_loss(o[1], o[2], labels, h.b_global, h.smooth_global)
            enc = enc + h.smooth_sent * gs
            out["gs_loss"] = gs.item()
        if h.enc_disc:
            rf = F.avg_pool2d(netd_forward(D, h, imgs).detach(), 4).view(B, -1)
            ff = F.avg_pool2d(feat, 4).view(B, -1)
            dl = contrastive_loss(rf, ff, labels, h.b_global, h.smooth_global)
            enc = enc + h.smooth_disc * dl
            out["disc_loss"] = dl.item()
        errG = errG_fake + enc
        gG = _grads(errG, G)
        optG.apply(PG, gG)
        out.update(errG_fake=errG_fake.item(), errG=errG.item(), grads_G=gG)
    return out
