"""cfg-driven XMC-GAN / DF-GAN trainer on MI355X -- drop-in for the reference entrypoint
(``python xmc_gan/train_gan.py --cfg xmc_gan/cfg/<preset>.yml [--gpu --seed --resume_epoch --log_type --bs --imsize]``).

Same flags, registries (``_GEN_ARCH`` / ``_DISC_ARCH``), helper names (``weight_init``, ``make_labels``,
``cosine_scores``, ``sent_loss``, ``img_loss``, ``train``, ``eval``) and the same arithmetic per iteration
as the reference loop (train_gan.py:174-293).  Every pass over an activation-, image- or parameter-sized tensor runs in a
hand-written HIP kernel through ``xmc_gan_amd``; what is left to ATen is batch-sized glue ([B], [B,cond] or [B,4,4,C]
tensors: the concatenations that build COND_DNET's input, slices of the padded logits, ``-mean(logit)``, the threshold
arithmetic of ``make_labels``) and the collectives.  Parity: ``--precision fp32`` reproduces the CPU reference to 1e-3
(measured ~1e-6); the default bf16 mode is checked against the quantisation-aware oracle and costs ~1e-2 on the losses
against the f32 reference (tests/test_models_gpu.py states both).  Extras that the reference lacks: ``--synthetic`` COCO-shaped random data
(the COCO pickles / DAMSM weights are not redistributable), ``--precision``, and one-process-per-GPU data
parallelism when launched under ``torch.distributed.run`` (gradient all-reduce + optional all-gathered
contrastive negatives, ``--gather_negatives``).
"""
import contextlib
import os
import sys

PROJ_DIR = os.path.abspath(os.path.join(os.path.dirname(os.path.realpath(__file__)), os.pardir))
if PROJ_DIR not in sys.path:
    sys.path.append(PROJ_DIR)

import argparse
import random
import time

import numpy as np
import torch

from xmc_gan.config.gan import cfg, cfg_from_file
from xmc_gan.model.df_gan import NetG as DF_GEN, NetD as DF_DISC
from xmc_gan.model.df_concept_gan import InNetG as CONCEPT_IN_DF_GEN, OutNetG as CONCEPT_OUT_DF_GEN, NetD as CONCEPT_NETD
from xmc_gan.dataset import SentTextDataset, WordTextDataset, test_transform, train_transform
from xmc_gan.model.concept_gan import InNetG as CONCEPT_INATTN_GEN, OutNetG as CONCEPT_OUTATTN_GEN
from xmc_gan.model.encoder import RNN_ENCODER, SBERT_ENCODER
from xmc_gan.utils.logger import setup_logger
from xmc_gan.utils.miscc import count_params
from xmc_gan.utils.visual import ScalarLog, fid_between, flush_saves, save_image, save_image_async, to_uint8_hwc
from xmc_gan_amd import ops, parallel
from xmc_gan_amd.optim import HipAdam

_GEN_ARCH = {"DF_GEN": DF_GEN, "CONCEPT_IN_DF_GEN": CONCEPT_IN_DF_GEN, "CONCEPT_OUT_DF_GEN": CONCEPT_OUT_DF_GEN,
             # word-attention generators of model/concept_gan.py; upstream keeps these two names commented out (train_gan.py:44)
             "CONCEPT_OUTATTN_GEN": CONCEPT_OUTATTN_GEN, "CONCEPT_INATTN_GEN": CONCEPT_INATTN_GEN}
_DISC_ARCH = {"DF_DISC": DF_DISC, "CONCEPT_NETD": CONCEPT_NETD}
_TEXT_DATASET = {"WORD": WordTextDataset, "SENT": SentTextDataset}
_TEXT_ARCH = {"RNN": RNN_ENCODER, "SBERT": SBERT_ENCODER}


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Train XMC-GAN')
    parser.add_argument('--cfg', type=str, default='xmc_gan/cfg/df_gan_sbert_seperate.yml')
    parser.add_argument('--gpu', dest='gpu_id', type=int, default=0)
    parser.add_argument('--seed', type=int, default=100)
    parser.add_argument('--resume_epoch', type=int, default=0)
    parser.add_argument('--log_type', type=str, default='tb')
    parser.add_argument('--bs', type=int, default=-1)
    parser.add_argument('--imsize', type=int, default=-1)
    # additions
    parser.add_argument('--synthetic', type=int, default=0, metavar='N',
                        help='train on N COCO-shaped random batches per epoch instead of data/<DATASET_NAME>')
    parser.add_argument('--max_epoch', type=int, default=-1)
    parser.add_argument('--data_dir', type=str, default='', help='dataset root (default: <repo>/data/<DATASET_NAME>)')
    parser.add_argument('--output_dir', type=str, default='', help='run directory (default: <repo>/output/<name>)')
    parser.add_argument('--precision', type=str, default=None, choices=['bf16', 'f16', 'fp32'])
    parser.add_argument('--gather_negatives', action='store_true',
                        help='data parallel: all-gather embeddings so the contrastive losses see world*batch negatives')
    parser.add_argument('--graph', type=int, default=1,
                        help='1 (default): replay the G+D iteration as hipGraphs (two eager warm-up iterations, then one capture per '
                             'N_CRITIC phase); 0: launch every kernel from Python (host-bound: ~55 ms per iteration whatever the batch)')
    parser.add_argument('--log_each_step', type=int, default=0,
                        help="1: the reference's loss line after EVERY generator step (one host synchronisation per iteration); default: "
                             'at the first step and every LOG_INTERVAL steps')
    return parser.parse_args(argv)


def weight_init(m):
    """Kaiming-normal (fan_in, relu gain) for every conv / linear weight, zero bias (train_gan.py:65-69)."""
    if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
        torch.nn.init.kaiming_normal_(m.weight, mode='fan_in', nonlinearity='relu')
        if m.bias is not None:
            torch.nn.init.constant_(m.bias, 0)


# --------------------------------------------------------------------------------------- contrastive head
def cosine_scores(emb0, emb1):
    """[bs,D] x [bs,D] -> [bs,bs] cosine similarities (train_gan.py:85-91)."""
    return ops.cosine_scores(emb0, emb1)


def make_labels(batch_size, sent_embs, b_global, p=0.6):
    """Positive-pair weights for the contrastive losses (train_gan.py:72-83)."""
    labels = torch.eye(batch_size, device=sent_embs.device)
    if b_global:
        sim_mat = cosine_scores(sent_embs, sent_embs)
        sim_mat.fill_diagonal_(3)
        global_pos = (sim_mat > p) & (sim_mat < 3)
        num_pos = global_pos.sum(1).clamp_(min=1) + 1
        global_weight = cfg.TRAIN.SMOOTH.GLOBAL if (cfg.TRAIN.SMOOTH.GLOBAL != 0.) else torch.reciprocal(num_pos.float())
        labels = (labels + global_weight * global_pos).clamp_(max=1)   # [bs] weight broadcasts along columns
    return labels.detach()


def _info_nce(a, b, labels, b_global):
    if not b_global:
        inv_num_pos, lab = None, None                   # labels are the identity: kernel fast path
    elif cfg.TRAIN.SMOOTH.GLOBAL == 0.:
        inv_num_pos, lab = torch.full((labels.size(0),), 0.5, device=labels.device), labels.contiguous()
    else:
        inv_num_pos, lab = torch.reciprocal((labels > 0).sum(1).float()), labels.contiguous()
    return ops.contrastive(a, b, lab, inv_num_pos)


def sent_loss(imgs, txts, labels, b_global):
    """image-embedding <-> sentence-embedding symmetric InfoNCE without temperature (train_gan.py:93-115)."""
    return _info_nce(imgs, txts, labels, b_global)


def img_loss(real_imgs, fake_imgs, labels, b_global):
    """real-feature <-> fake-feature symmetric InfoNCE (train_gan.py:117-139)."""
    return _info_nce(real_imgs, fake_imgs, labels, b_global)


# --------------------------------------------------------------------------------------- one iteration
class StepOptions:
    """Engine-side switches of an iteration (not part of the reference cfg)."""

    def __init__(self, gather_negatives=False, graph=False, log_each_step=True):
        self.gather_negatives = gather_negatives
        self.graph = graph                    # train(): replay the iteration as hipGraphs (xmc_gan_amd.graph.GraphedIteration)
        self.log_each_step = log_each_step    # train(): read the four logged losses back after every generator step


def _step(optimizer, scaler):
    """optimizer.step(), with the dynamic loss scale only where there is one (IEEE-half mode): any torch.optim optimizer works
    in the bf16 / fp32 modes, as upstream's torch.optim.Adam does"""
    return optimizer.step(scaler=scaler) if scaler is not None else optimizer.step()


def _set_requires_grad(module, flag):
    for p_ in module.parameters():
        p_.requires_grad_(flag)


def gan_iteration(netG, netD, optimizerG, optimizerD, imgs, sent_embs, words_embs, mask, noise, it_state,
                  opts=None):
    """One pass of the reference loop body (train_gan.py:185-291) on device tensors.

    Returns a dict of 0-d loss tensors (no host sync).  ``it_state['i']`` carries the N_CRITIC counter.
    """
    opts = opts or StepOptions()
    T, E = cfg.TRAIN, cfg.TRAIN.ENCODER_LOSS
    batch_size = mask.size(0)
    ops.new_iteration(imgs.device)                 # one memset for all weight-gradient scratch of this iteration
    gather = parallel.gather_rows if opts.gather_negatives else (lambda t: t)
    gather_all = parallel.gather_rows_multi if opts.gather_negatives else (lambda ts: list(ts))     # several tensors, one collective
    out = {}

    # ---- discriminator step (train_gan.py:187-229)
    psent_embs = sent_embs if cfg.DISC.SEPERATE else netG.proj_sent(sent_embs.float())
    nhwc_g = bool(getattr(netG, 'nhwc_out', False))             # generator can hand out its image in the engine layout
    batched = nhwc_g and isinstance(netD, DF_DISC) and not cfg.DISC.SPEC_NORM and ops.fused_blocks() and not ops.debug_switch('no_d2b')
    # the discriminator's 2B-image input of the batched pass below: the real images are converted into its first half, the
    # generator's last convolution writes its second half (no concatenation pass)
    both_h = (torch.empty((2 * batch_size, imgs.shape[2], imgs.shape[3], 8), dtype=ops.act_dtype(), device=imgs.device)
              if batched and getattr(netG, 'nhwc_dst_ok', False) else None)
    if nhwc_g:
        kw = dict(nhwc_dst=both_h[batch_size:]) if both_h is not None else {}
        fake, fake_h = netG(noise=noise, sent_embs=sent_embs, words_embs=words_embs, mask=mask, return_nhwc=True, **kw)
        assert both_h is None or fake_h.data_ptr() == both_h[batch_size:].data_ptr(), "generator ignored nhwc_dst"
    else:
        fake, fake_h = netG(noise=noise, sent_embs=sent_embs, words_embs=words_embs, mask=mask), None
    # converted once, used by both steps
    imgs_h = ops.to_nhwc8(imgs, out=both_h[:batch_size] if both_h is not None else None) if isinstance(netD, DF_DISC) else None
    # The discriminator does not couple the samples of a batch, so its passes over the real and the generated images
    # (193, 202) run as ONE pass over 2B images and the three COND_DNET calls (194, 203, 208) as one over 3B-1 rows: same
    # values, half the launches, twice the work per launch on the small maps at the end of D.  Not with spectral norm: there
    # every forward CALL advances the power iteration (modules.py:16-17), so the call pattern is part of the result.
    # data parallel: the trunk is handed out where it enters D's last blocks, for the two-part backward below (parallel.py)
    cuts = [] if (parallel.active() and isinstance(netD, DF_DISC) and netD.cut_block() is not None
                  and not ops.debug_switch('no_dp_overlap')) else None
    netD.cut_sink = cuts
    if imgs_h is not None and fake_h is not None and not cfg.DISC.SPEC_NORM and ops.fused_blocks() and not ops.debug_switch('no_d2b'):
        B = batch_size
        feats = netD(None, nhwc8=both_h if both_h is not None else torch.cat((imgs_h, fake_h.detach())))
        real_features, fake_features = feats[:B], feats[B:]
        ps_d = psent_embs.detach()
        rows = [feats, real_features[:B - 1]] if T.RMIS_LOSS else [feats]
        sents = [ps_d, ps_d, ps_d[1:B]] if T.RMIS_LOSS else [ps_d, ps_d]
        o_all = netD.COND_DNET(torch.cat(rows) if T.RMIS_LOSS else feats, sent_embs=torch.cat(sents))
        outputs_real = [o[:B] for o in o_all]
        errD_real = ops.hinge(o_all[0][:B], -1.0)
        errD_fake = ops.hinge(o_all[0][B:2 * B], 1.0)
        mis_loss = errD_fake
        if T.RMIS_LOSS:
            errD_mismatch = ops.hinge(o_all[0][2 * B:], 1.0)
            mis_loss = mis_loss + errD_mismatch
            out['errD_mismatch'] = errD_mismatch.detach()
    else:
        real_features = netD(imgs, nhwc8=imgs_h) if imgs_h is not None else netD(imgs)
        outputs_real = netD.COND_DNET(real_features, sent_embs=psent_embs.detach())
        errD_real = ops.hinge(outputs_real[0], -1.0)
        fake_features = netD(fake.detach(), nhwc8=fake_h.detach()) if fake_h is not None and imgs_h is not None else netD(fake.detach())
        outputs_fake = netD.COND_DNET(fake_features, sent_embs=psent_embs.detach())
        errD_fake = ops.hinge(outputs_fake[0], 1.0)
        mis_loss = errD_fake
        if T.RMIS_LOSS:
            outputs_mis = netD.COND_DNET(real_features[:(batch_size - 1)], sent_embs=psent_embs[1:batch_size].detach())
            errD_mismatch = ops.hinge(outputs_mis[0], 1.0)
            mis_loss = mis_loss + errD_mismatch
            out['errD_mismatch'] = errD_mismatch.detach()
    netD.cut_sink = None
    labels = None
    n_rows = batch_size * (parallel.world() if opts.gather_negatives else 1)
    enc_loss = 0.
    if E.SENT:
        assert cfg.DISC.SENT_MATCH or cfg.DISC.IMG_MATCH
        # (the sentence embeddings the labels are made from travel with the two embeddings the loss compares: one collective)
        g_sent, g_img, g_txt = gather_all((sent_embs.float(), outputs_real[1], outputs_real[2]))
        labels = make_labels(n_rows, sent_embs=g_sent, b_global=E.B_GLOBAL)
        ds_loss = sent_loss(imgs=g_img, txts=g_txt, labels=labels, b_global=E.B_GLOBAL)
        enc_loss = enc_loss + T.SMOOTH.SENT * ds_loss
        out['ds_loss'] = ds_loss.detach()
    elif E.WORD or E.DISC or E.VGG:
        labels = make_labels(n_rows, sent_embs=gather(sent_embs.float()), b_global=E.B_GLOBAL)
    if E.WORD:
        raise NotImplementedError
    errD = errD_real + (mis_loss * T.SMOOTH.MISMATCH) + enc_loss
    netG.zero_grad()
    netD.zero_grad()
    # IEEE-half mode only (ops.set_precision): each backward runs on a dynamic, device-resident scale x its loss (keeps 1/B-sized
    # gradients normal), HipAdam reads the f32 parameter gradients times 1 / scale and SKIPS the step on the device when one
    # of them is inf / NaN (after the all-reduce, so every rank decides alike).  None in the other modes.
    sc_d = ops.loss_scaler("D", imgs.device)
    loss_d = sc_d.scale(errD) if sc_d is not None else errD
    if cuts:
        # Backward in two parts.  D's head and last three blocks hold 93 % of its gradient bytes (75 of 81 MB at 256 px) and their backward
        # is the first quarter of the pass (small maps); the blocks before the cut hold 7 % and take the rest of the time (large maps).  The
        # first part's all-reduce is started as soon as its gradients exist and runs beside the second part; same gradients as one
        # `.backward()` (autograd executes the same nodes in the same order, in two calls).
        late, early = netD.late_parameters()
        late = [p_ for p_ in late if p_.requires_grad]
        gs = torch.autograd.grad(loss_d, cuts + late, allow_unused=True)
        for p_, g_ in zip(late, gs[len(cuts):]):
            p_.grad = g_
        pending = parallel.allreduce_mean_grads_begin(late)
        reached = [(c_, g_) for c_, g_ in zip(cuts, gs) if g_ is not None]          # (a handed-out tensor the loss does not use has no gradient)
        torch.autograd.backward([c_ for c_, _ in reached], [g_ for _, g_ in reached])
        parallel.allreduce_mean_grads_end(pending, early)
    else:
        loss_d.backward()
        parallel.allreduce_mean_grads(netD.parameters())
    _step(optimizerD, sc_d)
    out.update(errD=errD.detach(), errD_real=errD_real.detach(), errD_fake=errD_fake.detach())

    # ---- matching-aware gradient penalty on real pairs (train_gan.py:231-252)
    if T.MAGP:
        interpolated = imgs.detach().requires_grad_()
        sent_inter = psent_embs.detach().requires_grad_()
        # this forward's backward is differentiated again.  The fused discriminator blocks carry their own second-order node
        # (ops.ResDBwdFn); with spectral norm the blocks are composed from the fine-grained Functions as before
        # ops.second_order(): the blocks keep their residual branch (not just its sign bits) for the linearised forward
        with (ops.composable() if cfg.DISC.SPEC_NORM else contextlib.nullcontext()), ops.second_order():
            features = netD(interpolated)
            o = netD.COND_DNET(features, sent_inter)
        s_in = ops.gp_inner_scale()    # IEEE-half mode: the inner backward runs on s_in x ones, the penalty divides it out
        with ops.no_wgrad():           # first-order pass only needs d(logit)/d(inputs)
            grads = torch.autograd.grad(outputs=o[0], inputs=(interpolated, sent_inter),
                                        grad_outputs=torch.full_like(o[0], s_in), retain_graph=True, create_graph=True,
                                        only_inputs=True)
        # mean(||cat(grad0, grad1)||_2 ** 6) (241-247) in two passes over the 3*S*S-wide image gradient, no concatenation
        d_loss_gp = ops.grad_penalty(grads[0], grads[1], inner_scale=s_in)
        d_loss = 2.0 * d_loss_gp
        optimizerD.zero_grad()
        optimizerG.zero_grad()
        sc_gp = ops.loss_scaler("GP", imgs.device)
        (sc_gp.scale(d_loss) if sc_gp is not None else d_loss).backward()
        # the reference's autograd hands zero (not None) grads to every bias that feeds the logit
        # (their only path is through LeakyReLU'' == 0), which still advances Adam's moments/step.
        for name, p_ in netD.named_parameters():
            if p_.grad is None and name.endswith('.bias') and 'proj_match' not in name and _bias_on_logit_path(netD, name):
                p_.grad = torch.zeros_like(p_)
        parallel.allreduce_mean_grads(netD.parameters())
        _step(optimizerD, sc_gp)
        out['d_loss_gp'] = d_loss_gp.detach()

    # ---- generator step (train_gan.py:254-291)
    it_state['i'] = it_state.get('i', 0) + 1
    if it_state['i'] % T.N_CRITIC == 0:
        _set_requires_grad(netD, False)          # D's weight grads would be discarded (zero_grad at 226-227)
        try:
            features = netD(fake, nhwc8=fake_h) if fake_h is not None and imgs_h is not None else netD(fake)
            outputs = netD.COND_DNET(features, sent_embs=psent_embs)
            errG_fake = -outputs[0].float().mean()
            enc_loss = 0.0
            real_pooled = fake_pooled = None
            if E.DISC:
                with torch.no_grad():
                    real_again = netD(imgs, nhwc8=imgs_h) if imgs_h is not None else netD(imgs)
                    real_pooled = ops.global_avgpool(real_again.permute(0, 2, 3, 1).contiguous())
                fake_pooled = ops.global_avgpool(features.permute(0, 2, 3, 1).contiguous())
            # what the contrastive terms of this step compare across ranks, all-gathered in ONE collective (a seam of the captured iteration each)
            rows = gather_all(([outputs[1], outputs[2]] if E.SENT else []) + ([real_pooled, fake_pooled] if E.DISC else []))
            if E.SENT:
                gs_loss = sent_loss(imgs=rows[0], txts=rows[1], labels=labels, b_global=E.B_GLOBAL)
                enc_loss = enc_loss + T.SMOOTH.SENT * gs_loss
                out['gs_loss'] = gs_loss.detach()
            if E.WORD:
                raise NotImplementedError
            if E.DISC:
                disc_loss = img_loss(real_imgs=rows[-2], fake_imgs=rows[-1], labels=labels, b_global=E.B_GLOBAL)
                enc_loss = enc_loss + T.SMOOTH.DISC * disc_loss
                out['disc_loss'] = disc_loss.detach()
            if E.VGG:
                raise NotImplementedError
            errG = errG_fake + enc_loss
            netG.zero_grad()
            netD.zero_grad()
            sc_g = ops.loss_scaler("G", imgs.device)
            (sc_g.scale(errG) if sc_g is not None else errG).backward()
        finally:
            _set_requires_grad(netD, True)
        parallel.allreduce_mean_grads(netG.parameters())
        _step(optimizerG, sc_g)
        it_state['i'] = 0
        out.update(errG=errG.detach(), errG_fake=errG_fake.detach())
    out['fake'] = fake.detach()
    ops.end_iteration(imgs.device)                 # forward-only code between iterations (evaluation, sampling) stays out of the arena
    return out


def _bias_on_logit_path(netD, name):
    """conv_img.bias and the conv_s.bias of blocks whose learned shortcut is active (df_gan.py:286-288)."""
    if name == 'conv_img.bias':
        return True
    if name.startswith('downblocks.') and name.endswith('.conv_s.bias'):
        return netD.downblocks[int(name.split('.')[1])].learned_shortcut
    return False


# --------------------------------------------------------------------------------------- synthetic front end
class SyntheticCOCO:
    """COCO-shaped random batches with the tuple layout of the reference loader ``(imgs, [(caps, cap_lens)], keys)``
    (dataset.py:64): images uniform in [-1,1] like Normalize(0.5,0.5) output (dataset.py:34-37), captions as WordTextDataset
    pads them (int64 token ids in [1, V), zeros after the length; dataset.py:104-111)."""

    def __init__(self, n_batches, batch_size, img_size, max_len, seed, voca_size=27297, distinct=4):
        """``distinct``: batches i and i + distinct are the same tensors (all generated before the first one is handed out and kept in
        pinned host memory): drawing 50 M uniform numbers per 256 x 256 x 256 batch takes the host ~0.25 s, six times the iteration
        it feeds."""
        self.n, self.bs, self.size, self.max_len, self.seed, self.voca = n_batches, batch_size, img_size, max_len, seed, voca_size
        self.distinct = max(1, min(int(distinct), n_batches))
        self._cache = {}

    def __len__(self):
        return self.n

    def _batch(self, i):
        if i not in self._cache:
            g = torch.Generator().manual_seed(self.seed * 100003 + i)
            imgs = torch.rand(self.bs, 3, self.size, self.size, generator=g) * 2 - 1
            if torch.cuda.is_available():
                imgs = imgs.pin_memory()
            lens = torch.randint(5, self.max_len + 1, (self.bs,), generator=g)
            caps = torch.randint(1, self.voca, (self.bs, self.max_len), generator=g)
            caps = caps * (torch.arange(self.max_len)[None, :] < lens[:, None])
            self._cache[i] = (imgs, caps, lens)
        return self._cache[i]

    def __iter__(self):
        for i in range(self.distinct):
            self._batch(i)
        for i in range(self.n):
            imgs, caps, lens = self._batch(i % self.distinct)
            yield imgs, [(caps, lens)], [f'syn{i}_{j}' for j in range(self.bs)]


class SyntheticTextEncoder(torch.nn.Module):
    """Stands in for the frozen SBERT encoder (encoder.py:24-70; third-party package + checkpoint): returns
    words_embs [B,E,T], sent_embs [B,E], mask [B,T] (True = padding) as random tensors that are a function of the captions."""

    def __init__(self, emb_dim, max_len, seed, device):
        super().__init__()
        self.e, self.t, self.seed, self.device = emb_dim, max_len, seed, device

    def forward(self, caps, cap_lens):
        caps = torch.as_tensor(caps).cpu()
        g = torch.Generator().manual_seed(self.seed * 7919 + int(caps[0, :4].sum()))
        B = cap_lens.size(0)
        words = torch.randn(B, self.e, self.t, generator=g)
        sent = torch.randn(B, self.e, generator=g)
        mask = torch.arange(self.t)[None, :] >= cap_lens.cpu()[:, None]
        return words.to(self.device), sent.to(self.device), mask.to(self.device)


# --------------------------------------------------------------------------------------- epoch loop
_TIMED_FROM = 4          # train() reports images/s from the end of this step on: two eager warm-ups, the capture(s), one replay
def _log_epoch_scalars(writer, last, epoch):
    """the per-epoch scalars of the reference (train_gan.py:300-320): the losses of the epoch's last iteration"""
    if writer is None or 'errD' not in last:
        return
    writer.add_scalar('epoch', epoch, epoch)
    for tag, key in (('Loss_D', 'errD'), ('Loss_G', 'errG'), ('errD_real', 'errD_real'), ('errD_fake', 'errD_fake'),
                     ('errD_mismatch', 'errD_mismatch'), ('ds_loss', 'ds_loss'), ('gs_loss', 'gs_loss'), ('disc_loss', 'disc_loss')):
        if key in last:
            writer.add_scalar(tag, last[key].item(), epoch)
    writer.flush()


class _DeviceBatches:
    """The loader's batches with the image tensor already on its way to the device: while iteration i runs (a graph replay leaves the
    host idle), batch i+1 is fetched and uploaded on a side stream from pinned memory, so the 0.2 GB of a 256 x 256 x 256 f32 batch
    (~5 ms of PCIe time) is off the iteration's critical path.  Yields the loader's tuples with `imgs` on the device."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, device
        self.stream = torch.cuda.Stream(device=device) if device.type == 'cuda' else None

    def __len__(self):
        return len(self.loader)

    def _upload(self, data):
        imgs, texts_lst, keys = data
        if self.stream is None or not torch.is_tensor(imgs) or imgs.is_cuda:
            return (imgs.to(self.device) if torch.is_tensor(imgs) else imgs, texts_lst, keys), imgs, None
        host = imgs if imgs.is_pinned() else imgs.pin_memory()
        with torch.cuda.stream(self.stream):
            dev = host.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return (dev, texts_lst, keys), host, ev

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        while nxt is not None:
            (dev, texts_lst, keys), host, ev = nxt
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
                dev.record_stream(torch.cuda.current_stream())
            yield dev, texts_lst, keys, host
            try:                       # (after the consumer has launched its iteration: this upload overlaps it)
                nxt = self._upload(next(it))
            except StopIteration:
                nxt = None


def train(train_loader, test_loader, state_epoch, text_encoder, netG, netD, optimizerG, optimizerD, logger, model_dir,
          opts=None, img_dir=None, max_steps=None, writer=None):
    """Epoch loop with the reference's signature (train_gan.py:142); returns the last iteration's losses.

    With ``img_dir`` (rank 0) it keeps the reference's visual log: ``sents.txt`` and ``imgs.png`` of the first batch (146-160),
    ``fake_samples_<step>.png`` every LOG_INTERVAL steps (298-299), ``fake_samples_epoch_<epoch>.png`` from a fixed noise /
    caption batch in eval mode after every epoch (322-326); with ``writer`` the per-epoch scalars (300-320).

    ``opts.graph`` (the entry point's default, ``--graph 1``): the loop body runs as hipGraph replays
    (xmc_gan_amd.graph.GraphedIteration: two eager warm-up iterations, one capture per N_CRITIC phase, collectives as eager
    seams between graph segments); a batch is copied into the graph's static inputs and the losses stay on the device until they
    are logged -- at the first step and every LOG_INTERVAL steps unless ``opts.log_each_step``."""
    device = next(netG.parameters()).device
    opts = opts or StepOptions()
    it_state, last, nsteps = {}, {}, 0
    fixed = None
    visual = img_dir is not None and parallel.rank() == 0
    graphed, use_graph = None, bool(opts.graph) and device.type == 'cuda'
    batches = _DeviceBatches(train_loader, device)
    t_mark, thr = [None, 0], None

    def step_fn(imgs_, sent_, words_, mask_, noise_, st_):
        return gan_iteration(netG, netD, optimizerG, optimizerD, imgs_, sent_, words_, mask_, noise_, st_, opts)

    for epoch in range(state_epoch + 1, cfg.TRAIN.MAX_EPOCH + 1):
        netG.train()
        netD.train()
        sampler = getattr(train_loader, 'sampler', None)
        if isinstance(sampler, torch.utils.data.distributed.DistributedSampler):
            sampler.set_epoch(epoch)             # data parallel: a new permutation (and new per-rank shards) every epoch
        for step, data in enumerate(batches):
            imgs, texts_lst, keys, imgs_host = data
            caps, cap_lens = texts_lst[0]
            with torch.no_grad():
                words_embs, sent_embs, mask = text_encoder(caps, cap_lens)
            words_embs, sent_embs = words_embs.detach(), sent_embs.detach()
            if visual and fixed is None:         # the reference takes these from the loader's first batch before the loop
                i2w = getattr(getattr(train_loader, 'dataset', None), 'i2w', None)
                with open(f'{img_dir}/sents.txt', 'w') as f:
                    if torch.is_tensor(caps):            # WORD captions: token ids (index_to_sent, dataset.py) or the ids themselves
                        for row, n in zip(caps.tolist(), torch.as_tensor(cap_lens).tolist()):
                            f.write((' '.join(i2w[t] for t in row[:n]) if i2w else ' '.join(str(t) for t in row[:n])) + ' \n')
                    else:                                # SENT captions: the sentences
                        for sent in caps:
                            f.write(f'{sent} \n')
                fixed = dict(noise=torch.randn(mask.size(0), cfg.TRAIN.NOISE_DIM).to(device), sent=sent_embs.clone(),
                             words=words_embs.clone(), mask=mask.clone())
                save_image(imgs_host, f'{img_dir}/imgs.png', normalize=True, scale_each=True)
            noise = torch.randn(mask.size(0), cfg.TRAIN.NOISE_DIM).to(device, non_blocking=True)   # CPU generator, as upstream (197-198)
            inputs = (imgs, sent_embs, words_embs, mask, noise)
            if use_graph and graphed is None:
                from xmc_gan_amd.graph import GraphedIteration
                graphed = GraphedIteration(step_fn, inputs, n_critic=cfg.TRAIN.N_CRITIC, warmup=2)
                graphed.it_state = it_state
            if graphed is not None and all(s_.shape == t_.shape and s_.dtype == t_.dtype for s_, t_ in zip(graphed.static_in, inputs)):
                try:
                    last = dict(graphed(*inputs))
                except Exception as e:           # noqa: BLE001 -- a capture that failed (an operator that synchronises, memory)
                    if parallel.world() > 1:
                        raise                    # the ranks' collectives are out of step: no in-band way to agree on a fallback
                    logger.info(f'hipGraph capture failed ({type(e).__name__}: {e}); continuing with eager launches')
                    it_state = dict(graphed.it_state)
                    graphed.close()
                    graphed, use_graph = None, False
                    last = step_fn(*inputs, it_state)
            else:                                # eager: --graph 0, or a batch of another shape (a loader without drop_last)
                last = step_fn(*inputs, it_state)
            first = nsteps == 0
            if device.type == 'cuda' and t_mark[0] is None and nsteps == _TIMED_FROM + 2 * (cfg.TRAIN.N_CRITIC - 1):
                torch.cuda.synchronize(device)           # the run's throughput: from the end of step _TIMED_FROM (warm-ups and captures done) ...
                t_mark[0], t_mark[1] = time.perf_counter(), nsteps + 1
            if 'errG' in last and (opts.log_each_step or first or (step + 1) % cfg.TRAIN.LOG_INTERVAL == 0):
                logger.info(f'[{epoch}/{cfg.TRAIN.MAX_EPOCH}][{step + 1}/{len(train_loader)}] '
                            f'Loss_D: {last["errD"].item():.3f} Loss_G: {last["errG"].item():.3f} '
                            f'errD_real: {last["errD_real"].item():.3f} errD_fake: {last["errD_fake"].item():.3f} ')
            if visual and (step + 1) % cfg.TRAIN.LOG_INTERVAL == 0:
                # (copied to the host now, encoded on the writer thread: utils/visual.py)
                save_image_async(last['fake'], f'{img_dir}/fake_samples_{step + 1:03d}.png', normalize=True, scale_each=True)
            nsteps += 1
            if max_steps is not None and nsteps >= max_steps:
                flush_saves()
                return _detached(last)
        if t_mark[0] is not None and nsteps > t_mark[1]:
            torch.cuda.synchronize(device)               # ... to the end of the epoch (one synchronisation per epoch)
            dt_, n_ = time.perf_counter() - t_mark[0], nsteps - t_mark[1]
            thr = dict(steps=n_, seconds=round(dt_, 4), ms_per_step=round(1e3 * dt_ / n_, 3),
                       images_per_s=round(parallel.world() * cfg.TRAIN.BATCH_SIZE * n_ / dt_, 1), hipgraph=graphed is not None)
            logger.info(f'throughput: {thr["images_per_s"]} images/s ({thr["ms_per_step"]} ms per iteration over {n_} iterations, '
                        f'{"hipGraph replay" if graphed is not None else "eager launches"})')
            t_mark[0], t_mark[1] = time.perf_counter(), nsteps
        if parallel.rank() == 0:
            _log_epoch_scalars(writer, last, epoch)
        # IEEE-half mode: the dynamic loss scales and the optimizer steps the found-inf check skipped so far (one host read per epoch)
        sc_stats = ops.loss_scaler_stats()
        if sc_stats:
            last['loss_scale'] = sc_stats
            logger.info('loss scale ' + ' '.join(f"{k}: {v['scale']:g} ({v['skipped_steps']} skipped)" for k, v in sc_stats.items()))
            if all(v['scale'] <= 1.0 and v['last_step_skipped'] for v in sc_stats.values()):
                raise FloatingPointError('every backward of the f16 mode overflows at loss scale 1: the run has diverged')
        if visual and fixed is not None:
            with torch.no_grad():
                netG.eval()
                fake = netG(noise=fixed['noise'], sent_embs=fixed['sent'], words_embs=fixed['words'], mask=fixed['mask'])
                save_image_async(fake, f'{img_dir}/fake_samples_epoch_{epoch:03d}.png', normalize=True, scale_each=True)
        if epoch > 50 and parallel.rank() == 0:
            torch.save(netG.state_dict(), f'{model_dir}/netG_{epoch:03d}.pth')
            torch.save(netD.state_dict(), f'{model_dir}/netD_{epoch:03d}.pth')
            torch.save(optimizerG.state_dict(), f'{model_dir}/optimizerG.pth')
            torch.save(optimizerD.state_dict(), f'{model_dir}/optimizerD.pth')
            if ops.loss_scaler_state():          # IEEE-half mode: the dynamic loss scales are optimizer state too
                torch.save(ops.loss_scaler_state(), f'{model_dir}/loss_scale.pth')
            logger.info('Save models')
            if test_loader is not None:
                eval(loader=test_loader, state_epoch=epoch, text_encoder=text_encoder, netG=netG, logger=logger, num_samples=6000,
                     save_dir=f'{img_dir}/test' if img_dir else None, org_dir=f'{img_dir}/org' if img_dir else None, writer=writer)
    flush_saves()                    # the sample grids queued for the writer thread are on disk when train() returns
    last = _detached(last)
    if thr is not None:
        last['throughput'] = thr
    if graphed is not None:
        last['hipgraph'] = True
        graphed.close()
    return last


def _detached(last):
    """the last iteration's losses as tensors of their own (a graph's static outputs are overwritten by the next replay and freed with it)"""
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in last.items()}


@torch.no_grad()
def eval(loader, state_epoch, text_encoder, netG, logger, num_samples=6000, save_dir=None, org_dir=None, writer=None):
    """Generate images for the test loader and score them (train_gan.py:338-395): every generated image goes to
    ``save_dir/<key>.png`` and, unless ``org_dir`` already holds ``num_samples`` files, every real one to ``org_dir/<key>.png``,
    as 8-bit PNGs of (x + 1) * 127.5; FID between the two directories when ``pytorch_fid`` is importable (logged and written to
    the scalar log as 'FID').  Returns (uint8 tensor of the generated images, FID or None)."""
    from PIL import Image
    netG.eval()
    device = next(netG.parameters()).device
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
    save_org = False
    if org_dir:
        os.makedirs(org_dir, exist_ok=True)
        save_org = len(os.listdir(org_dir)) != num_samples
    cnt, outs = 0, []
    for imgs, texts_lst, keys in loader:
        caps, cap_lens = texts_lst[0]
        words_embs, sent_embs, mask = text_encoder(caps, cap_lens)
        noise = torch.randn(sent_embs.size(0), cfg.TRAIN.NOISE_DIM).to(device)
        fake = netG(noise=noise, sent_embs=sent_embs, words_embs=words_embs, mask=mask)
        outs.append(((fake + 1.0) * 127.5).clamp(0, 255).to(torch.uint8).cpu())
        for j in range(fake.size(0)):
            if save_dir:
                Image.fromarray(to_uint8_hwc(fake[j])).save(f'{save_dir}/{keys[j]}.png')
            if save_org:
                Image.fromarray(to_uint8_hwc(imgs[j])).save(f'{org_dir}/{keys[j]}.png')
        cnt += fake.size(0)
        if cnt >= num_samples:
            break
    fid = fid_between(org_dir, save_dir, device) if (save_dir and org_dir and cnt) else None
    if fid is None:
        logger.info(f' epoch {state_epoch}, generated {cnt} images (FID not computed: pytorch_fid is not installed)')
    else:
        logger.info(f' epoch {state_epoch}, FID : {fid}')
        if writer is not None:
            writer.add_scalar('FID', fid, state_epoch)
    return (torch.cat(outs) if outs else None), fid


def build_models(device):
    """netG / netD from the registries + the two Adam optimizers (train_gan.py:470-484)."""
    netG = _GEN_ARCH[cfg.GEN.ENCODER_NAME](cfg).to(device)
    netD = _DISC_ARCH[cfg.DISC.ENCODER_NAME](cfg, is_disc=True).to(device)
    if cfg.TRAIN.HE_INIT:
        netG.apply(weight_init)
        netD.apply(weight_init)
    O = cfg.TRAIN.OPT
    optimizerG = HipAdam(netG.parameters(), lr=O.G_LR, betas=(O.G_BETA1, O.G_BETA2))
    optimizerD = HipAdam(netD.parameters(), lr=O.D_LR, betas=(O.D_BETA1, O.D_BETA2))
    return netG, netD, optimizerG, optimizerD


def main(argv=None):
    args = parse_args(argv)
    cfg_from_file(args.cfg)
    if args.imsize != -1:
        cfg.IMG.SIZE = args.imsize
    if args.bs != -1:
        cfg.TRAIN.BATCH_SIZE = args.bs
    if args.max_epoch != -1:
        cfg.TRAIN.MAX_EPOCH = args.max_epoch
    if args.precision:
        ops.set_precision(args.precision)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(args.gpu_id)))
    if not torch.cuda.is_available():
        raise RuntimeError('xmc_gan/train_gan.py needs an MI355X (HIP kernels only; the CPU restatement lives in oracle/)')
    # one process per GPU over RCCL ('nccl'); XMC_DIST_BACKEND=gloo is the one-card rehearsal of the same code (ranks share card 0 when
    # there are fewer cards than ranks; tests/test_parallel_gpu.py)
    backend = os.environ.get('XMC_DIST_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        torch.distributed.init_process_group(backend)
    rank = parallel.rank()

    seed = args.seed + rank
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)

    output_dir = args.output_dir or f'{PROJ_DIR}/output/{cfg.DATASET_NAME}{cfg.IMG.SIZE}_{cfg.CONFIG_NAME}_{args.seed}'
    img_dir, log_dir, model_dir = output_dir + '/img', output_dir + '/log', output_dir + '/model'
    if rank == 0:
        for d_ in (output_dir, img_dir, log_dir, model_dir):
            os.makedirs(d_, exist_ok=True)
    logger = setup_logger(name=cfg.CONFIG_NAME, save_dir=log_dir if rank == 0 else None, distributed_rank=rank)
    logger.info('Using config:')
    logger.info(cfg)
    logger.info(f'seed now is : {args.seed}')

    test_loader = None
    if args.synthetic > 0:
        train_loader = SyntheticCOCO(args.synthetic, cfg.TRAIN.BATCH_SIZE, cfg.IMG.SIZE, cfg.TEXT.MAX_LENGTH, seed,
                                     cfg.TEXT.VOCA_SIZE)
    else:       # the reference's loaders (train_gan.py:440-457); each rank draws its own shuffled batches
        data_dir = args.data_dir or f'{PROJ_DIR}/data/{cfg.DATASET_NAME}'
        data_arch = _TEXT_DATASET[cfg.TEXT.TYPE]
        train_set = data_arch(data_dir=data_dir, mode='train', transform=train_transform(cfg.IMG.SIZE), cfg=cfg)
        test_set = data_arch(data_dir=data_dir, mode='test', transform=test_transform(cfg.IMG.SIZE), cfg=cfg)
        sampler = torch.utils.data.distributed.DistributedSampler(train_set, world, rank, shuffle=True, drop_last=True) \
            if world > 1 else None
        train_loader = torch.utils.data.DataLoader(train_set, batch_size=cfg.TRAIN.BATCH_SIZE, drop_last=True,
                                                   shuffle=sampler is None, sampler=sampler,
                                                   num_workers=int(cfg.TRAIN.NUM_WORKERS), pin_memory=True)
        test_loader = torch.utils.data.DataLoader(test_set, batch_size=cfg.TRAIN.BATCH_SIZE, drop_last=True, shuffle=False,
                                                  num_workers=int(cfg.TRAIN.NUM_WORKERS))
    if args.synthetic > 0 and cfg.TEXT.ENCODER_NAME != 'RNN':
        text_encoder = SyntheticTextEncoder(cfg.TEXT.EMBEDDING_DIM, cfg.TEXT.MAX_LENGTH, seed, device)
    else:       # train_gan.py:459-468
        text_encoder = _TEXT_ARCH[cfg.TEXT.ENCODER_NAME](cfg=cfg).to(device)
        enc_path = f'{PROJ_DIR}/{cfg.TEXT.ENCODER_DIR}'
        if cfg.TEXT.ENCODER_DIR != '' and (args.synthetic <= 0 or os.path.isfile(enc_path)):
            text_encoder.load_state_dict(torch.load(enc_path, map_location=device))
        elif world > 1:     # random (synthetic-run) encoder weights must agree across ranks
            for p_ in text_encoder.parameters():
                torch.distributed.broadcast(p_.data, 0)
        for p_ in text_encoder.parameters():
            p_.requires_grad = False
        text_encoder.eval()

    netG, netD, optimizerG, optimizerD = build_models(device)
    if world > 1:   # same initial weights (and spectral-norm u/v vectors, which then evolve identically) on every rank
        for p_ in list(netG.parameters()) + list(netD.parameters()) + list(netG.buffers()) + list(netD.buffers()):
            torch.distributed.broadcast(p_.data, 0)
    logger.info(f'netG # of parameters: {count_params(netG)}')
    logger.info(f'netD # of parameters: {count_params(netD)}')

    state_epoch = args.resume_epoch
    if state_epoch != 0:
        netG.load_state_dict(torch.load(f'{model_dir}/netG_{state_epoch:03d}.pth', map_location=device))
        netD.load_state_dict(torch.load(f'{model_dir}/netD_{state_epoch:03d}.pth', map_location=device))
        optimizerG.load_state_dict(torch.load(f'{model_dir}/optimizerG.pth', map_location=device))
        optimizerD.load_state_dict(torch.load(f'{model_dir}/optimizerD.pth', map_location=device))
        if os.path.isfile(f'{model_dir}/loss_scale.pth'):
            ops.load_loss_scaler_state(torch.load(f'{model_dir}/loss_scale.pth', map_location='cpu'), device)
        logger.info(f'Load models, epoch : {state_epoch}')
    elif cfg.DISC.ENCODER_DIR:
        netD.load_state_dict(torch.load(f'{PROJ_DIR}/{cfg.DISC.ENCODER_DIR}', map_location=device), strict=False)

    writer = ScalarLog(log_dir, args.log_type, run_name=cfg.CONFIG_NAME) if rank == 0 else None
    last = train(train_loader=train_loader, test_loader=test_loader, state_epoch=state_epoch, text_encoder=text_encoder,
                 netG=netG, netD=netD, optimizerG=optimizerG, optimizerD=optimizerD, logger=logger, model_dir=model_dir,
                 opts=StepOptions(gather_negatives=args.gather_negatives, graph=bool(args.graph), log_each_step=bool(args.log_each_step)),
                 img_dir=img_dir if rank == 0 else None, writer=writer)
    if writer is not None:
        writer.close()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.destroy_process_group()
    main.last_models = (netG, netD)          # (for callers that drive main() in-process: the tests compare runs by their final weights)
    return last


if __name__ == '__main__':
    main()
