"""diagnostic (not a test): where does the bf16 engine leave the quantisation-aware oracle?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import xmc_ref as X
from xmc_gan_amd import ops
from parity_util import DEV, build_product, rel_err, mean_abs_err, setup_cfg

ops.set_precision("bf16")
over = {"TRAIN.NCH": int(sys.argv[1])} if len(sys.argv) > 1 else {"TRAIN.NCH": 8}
cfg, h = setup_cfg("df_gan_damsm_nomagp.yml", **over)
PG, PD = X.synth_params(X.gen_shapes(h), 5), X.synth_params(X.netd_shapes(h), 6)
b = X.synth_batch(h, 4, seed=200, words_len=20)
netG, netD, _, _ = build_product(h, PG, PD)
with torch.no_grad():
    f32 = X.gen_forward(PG, h, b["noise"], b["sent_embs"])
    with X.quant(True):
        fq = X.gen_forward(PG, h, b["noise"], b["sent_embs"])
        psq = X.proj_sent(PG, b["sent_embs"])
        dq = X.netd_forward(PD, h, b["imgs"])
        lq = X.cond_dnet(PD, h, dq, psq)
        dfq = X.netd_forward(PD, h, fq)
        lfq = X.cond_dnet(PD, h, dfq, psq)
    d32 = X.netd_forward(PD, h, b["imgs"])
    l32 = X.cond_dnet(PD, h, d32, X.proj_sent(PG, b["sent_embs"]))
    fp = netG(noise=b["noise"].to(DEV), sent_embs=b["sent_embs"].to(DEV))
    ps = netG.proj_sent(b["sent_embs"].to(DEV))
    dp = netD(b["imgs"].to(DEV))
    lp = netD.COND_DNET(dp, sent_embs=ps)
    dfp = netD(fq.to(DEV))                       # D on the ORACLE's fake image: isolates D
    lfp = netD.COND_DNET(dfp, sent_embs=ps)
print("G image   : vs quant", mean_abs_err(fp, fq), " vs f32", mean_abs_err(fp, f32), " (quant vs f32:", mean_abs_err(fq, f32), ")")
print("D feat    : vs quant", rel_err(dp, dq), " vs f32", rel_err(dp, d32))
print("logit real: vs quant", rel_err(lp[0], lq[0]), " vs f32", rel_err(lp[0], l32[0]), lp[0].flatten().tolist(), lq[0].flatten().tolist())
print("img emb   : vs quant", rel_err(lp[1], lq[1]))
print("D(fake_q) feat vs quant", rel_err(dfp, dfq), " logit", rel_err(lfp[0], lfq[0]))
# block by block through D on the real image
a = X.disc_arch(h.img_size, h.nch)
import torch.nn.functional as F
with torch.no_grad(), X.quant(True):
    x = b["imgs"]
    out = X.q(F.conv2d(X.q(x), X.qw(PD["conv_img.weight"]), PD["conv_img.bias"], 1, 1))
    xo = netD.conv_img(ops.to_nhwc8(x.to(DEV)))
    print("conv_img", rel_err(xo.permute(0, 3, 1, 2)[:, : out.size(1)], out))
    for i, blk in enumerate(netD.downblocks):
        p = f"downblocks.{i}"
        r = X.q(F.leaky_relu(F.conv2d(out, X.qw(PD[f"{p}.conv_r.0.weight"]), None, 2, 1), 0.2))
        r2 = X.q(F.leaky_relu(F.conv2d(r, X.qw(PD[f"{p}.conv_r.2.weight"]), None, 1, 1), 0.2))
        s = X.q(F.avg_pool2d(out, 2))
        if f"{p}.conv_s.weight" in PD and a["cin"][i + 1] != a["cout"][i + 1]:
            s = X.q(F.conv2d(s, X.qw(PD[f"{p}.conv_s.weight"]), PD[f"{p}.conv_s.bias"]))
        out = X.q(s + PD[f"{p}.gamma"] * r2)
        # product block on the ORACLE's input of this block
        xin = (X.q(F.conv2d(X.q(x), X.qw(PD["conv_img.weight"]), PD["conv_img.bias"], 1, 1)) if i == 0 else prev_out)
        xi = xin.permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16)
        po = blk(xi)
        print(f"block {i}: product(block | oracle input) vs oracle", rel_err(po.permute(0, 3, 1, 2), out), " exact-equal frac",
              (po.permute(0, 3, 1, 2).float().cpu() == out).float().mean().item())
        prev_out = out
