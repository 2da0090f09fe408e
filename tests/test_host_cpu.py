"""CPU-only checks of the host side: C-ABI library exports, cfg schema/merge rules, module state_dict parity with the
oracle's key tables (pinned to the reference by the golden test), preset loading."""
import copy
import ctypes
import glob
import os
import re
import sys

import json

import numpy as np
import pytest
import torch

import xmc_ref as X
from golden_util import CFG_DIR, ROOT


@pytest.mark.parametrize("variant", ["bf16", "f16"])
def test_library_loads_and_exports_every_declared_symbol(variant):
    """both builds of the kernels (bf16 and IEEE-half storage: the same sources, common.h) export the whole C ABI"""
    import xmc_gan_amd.lib as L
    lib = L.load(variant)
    assert lib.xmc_half_format() == (0 if variant == "bf16" else 1)
    hdr = open(os.path.join(ROOT, "include", "xmc_gan_hip.h")).read()
    declared = set(re.findall(r"\b(xmc_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/xmc_gan_hip.h but not exported"
    assert set(L.EXPORTS) == declared
    assert lib.xmc_abi_version() == L.ABI_VERSION == 12
    assert lib.xmc_adam_step_scaled(None, 1, None, 1, 0.0, 0.0, 0.0, 0.0, None, None, 7, 2.0, 0.5, 1, None) == -1
    # argument validation happens before any launch, so it is safe without a GPU
    d = L.ConvDesc()
    assert lib.xmc_conv_igemm(ctypes.byref(d), None) == -1
    assert lib.xmc_conv_wgrad(ctypes.byref(d), None, None) == -1
    assert ctypes.sizeof(L.ConvDesc) == 440 and ctypes.sizeof(L.AdamEntry) == 48 and ctypes.sizeof(L.PackJob) == 64   # = the C structs
    # the entry points added for the callers either side of the step reject bad arguments the same way (nothing launched)
    import numpy as np
    assert np.dtype(L.GEMM_PROBLEM).itemsize == 88                    # sizeof(XmcGemmProblem)
    one = ctypes.c_void_p(8)                                           # any non-NULL value; never dereferenced on these paths
    assert lib.xmc_gemm_group(None, 1, None) == -1 and lib.xmc_gemm_group(one, 0, None) == -1
    tab = np.zeros(1, dtype=L.GEMM_PROBLEM)                            # NULL operand pointers inside the table
    assert lib.xmc_gemm_group(ctypes.c_void_p(tab.ctypes.data), 1, None) == -1
    assert lib.xmc_embedding_gather(one, one, one, 4, 302, 10, None) == -2        # width not a multiple of 4 floats
    assert lib.xmc_embedding_gather(None, one, one, 4, 300, 10, None) == -1
    assert lib.xmc_lstm_bidir(one, one, one, one, one, 4, 20, 64, None) == -1     # only H = 128 is built
    assert lib.xmc_gru_bidir(one, one, one, one, one, one, 4, 20, 64, None) == -1
    assert lib.xmc_signmask_apply(one, None, one, 64, 0.2, 0, None) == -1 and lib.xmc_signmask_apply(one, one, one, 12, 0.2, 0, None) == -2
    assert lib.xmc_conv_pw1x1_masked_src(ctypes.byref(L.ConvDesc()), one, one, 0.2, None) == -1      # null tensors in the descriptor
    assert lib.xmc_conv_pw1x1_split(ctypes.byref(L.ConvDesc()), None) == -1
    assert lib.xmc_spectral_sigma(None, one, one, one, one, None, 8, 8, 1, 1e-12, None) == -1
    assert lib.xmc_spectral_bwd(one, one, one, one, one, one, None, 8, 8, None) == -1
    assert lib.xmc_affine2_act_fwd(one, one, one, None, None, one, 1, 16, 12, 0.0, 0, None) == -2   # channels % 8


def test_missing_library_fails_loudly(monkeypatch):
    import xmc_gan_amd.lib as L
    monkeypatch.setattr(L, "_libs", {})
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libxmc_gan_hip.so")
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        L.load()


def test_ops_refuse_cpu_tensors():
    from xmc_gan_amd import ops
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.conv2d(torch.zeros(1, 4, 4, 8), torch.nn.Parameter(torch.zeros(8, 8, 3, 3)), None, ops.ConvGeom(8, 8, 3, 1, 1))


def test_cfg_merge_rules():
    from xmc_gan.config import gan
    gan.reset_cfg()
    with pytest.raises(KeyError):
        gan._merge_a_into_b(gan.AttrDict({"NOPE": 1}), gan.cfg)
    with pytest.raises(ValueError):
        gan._merge_a_into_b(gan.AttrDict({"TRAIN": {"NCH": "32"}}), gan.cfg)
    gan._merge_a_into_b(gan.AttrDict({"TRAIN": {"NCH": 16, "SMOOTH": {"SENT": 0.25}}}), gan.cfg)
    assert gan.cfg.TRAIN.NCH == 16 and gan.cfg.TRAIN.SMOOTH.SENT == 0.25 and gan.cfg.TRAIN.SMOOTH.DISC == 1.0
    gan.reset_cfg()
    assert gan.cfg.TRAIN.NCH == 32 and gan.cfg.DISC.SPEC_NORM is True


@pytest.mark.parametrize("yml", sorted(os.path.basename(f) for f in glob.glob(os.path.join(CFG_DIR, "*.yml"))))
def test_presets_load_and_modules_match_reference_state_dict(yml):
    from xmc_gan.config import gan
    import xmc_gan.train_gan as tg
    gan.reset_cfg()
    gan.cfg_from_file(os.path.join(CFG_DIR, yml))
    cfg = gan.cfg
    h = X.Hyper.from_cfg(cfg)
    if cfg.DISC.ENCODER_NAME == "CONCEPT_NETD":
        with pytest.raises(NotImplementedError):
            tg._DISC_ARCH[cfg.DISC.ENCODER_NAME](cfg, is_disc=True)        # as upstream (df_concept_gan.py:587)
        return
    for size in (64, 128, 256):
        cfg.IMG.SIZE = size
        h.img_size = size
        netD = tg._DISC_ARCH[cfg.DISC.ENCODER_NAME](cfg, is_disc=True)
        assert {k: tuple(v.shape) for k, v in netD.state_dict().items()} == X.netd_shapes(h)
        try:
            netG = tg._GEN_ARCH[cfg.GEN.ENCODER_NAME](cfg)
        except NotImplementedError:
            continue
        assert {k: tuple(v.shape) for k, v in netG.state_dict().items()} == X.gen_shapes(h)
        # weight_init reaches every conv/linear exactly like upstream's isinstance checks
        n_init = sum(isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)) for m in netG.modules())
        assert n_init == sum(1 for k in X.gen_shapes(h) if k.endswith(".weight") and ".gn" not in k and "proj_edge" not in k) \
            or cfg.GEN.ENCODER_NAME != "DF_GEN"


def test_spectral_norm_layers_mirror_the_legacy_hook_state():
    """DISC.SPEC_NORM=True: parameters/buffers named and shaped as torch.nn.utils.spectral_norm leaves them
    (model/modules.py:16-17,31-32), u/v drawn in its RNG order, ``weight`` a plain attribute."""
    from xmc_gan.config import gan
    import xmc_gan.train_gan as tg
    from xmc_gan.model.modules import conv2d_nxn, linear
    gan.reset_cfg()
    gan.cfg_from_file(os.path.join(CFG_DIR, "df_gan_damsm.yml"))
    gan.cfg.DISC.SPEC_NORM = True
    h = X.Hyper.from_cfg(gan.cfg)
    assert h.spec_norm
    netD = tg._DISC_ARCH["DF_DISC"](gan.cfg, is_disc=True)
    assert {k: tuple(v.shape) for k, v in netD.state_dict().items()} == X.netd_shapes(h)
    names = {n for n, _ in netD.named_parameters()}
    assert "conv_img.weight_orig" in names and not any(n.endswith(".weight") for n in names)
    assert {n for n, _ in netD.named_buffers()} == {k for k in X.netd_shapes(h) if k.endswith(("_u", "_v"))}
    for mk, ref in ((lambda: conv2d_nxn(5, 7, 3, 1, 1, spec_norm=True), lambda: torch.nn.Conv2d(5, 7, 3, 1, 1)),
                    (lambda: linear(6, 4, spec_norm=True), lambda: torch.nn.Linear(6, 4))):
        torch.manual_seed(11)
        mine = mk()
        torch.manual_seed(11)
        theirs = torch.nn.utils.spectral_norm(ref())
        sd_m, sd_t = mine.state_dict(), theirs.state_dict()
        assert list(sd_m.keys()) == list(sd_t.keys())
        for k in sd_m:
            assert torch.equal(sd_m[k], sd_t[k]), k
        assert not isinstance(mine.weight, torch.nn.Parameter)
        assert mine.weight.data_ptr() == mine.weight_orig.data_ptr()        # weight_init reaches weight_orig until .cuda()
        theirs.load_state_dict(sd_m)
        mine.load_state_dict(sd_t)


def test_adam_state_dict_interchanges_with_torch_adam():
    """optimizerG.pth / optimizerD.pth written by the reference (torch.optim.Adam, train_gan.py:331-332) load into HipAdam
    and vice versa (resume path, train_gan.py:492-493).  Host logic only: no step is taken here."""
    from xmc_gan_amd.optim import HipAdam
    torch.manual_seed(0)
    ps_t = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2))]
    adam = torch.optim.Adam(ps_t, lr=4e-4, betas=(0.0, 0.9))
    for _ in range(3):
        for p in ps_t[:2]:                      # the third parameter never gets a gradient: no state, as upstream
            p.grad = torch.randn_like(p)
        adam.step()
    sd_t = copy.deepcopy(adam.state_dict())     # as after torch.load: load_state_dict does not copy the state tensors
    ps_h = [torch.nn.Parameter(p.detach().clone()) for p in ps_t]
    hip = HipAdam(ps_h, lr=1e-3, betas=(0.5, 0.5))
    hip.load_state_dict(sd_t)
    assert hip.param_groups[0]["lr"] == 4e-4 and tuple(hip.param_groups[0]["betas"]) == (0.0, 0.9)
    assert set(hip.state.keys()) == set(ps_h[:2])
    for p_h, p_t in zip(ps_h[:2], ps_t[:2]):
        st = hip.state[p_h]
        assert st["step"].dtype == torch.int32 and st["step"].shape == (1,) and int(st["step"]) == 3
        assert torch.equal(st["exp_avg"], adam.state[p_t]["exp_avg"]) and torch.equal(st["exp_avg_sq"], adam.state[p_t]["exp_avg_sq"])
    sd_h = copy.deepcopy(hip.state_dict())
    assert sd_h["state"].keys() == sd_t["state"].keys()
    for k in sd_t["state"]:
        assert sd_h["state"][k]["step"].dtype == torch.float32 and sd_h["state"][k]["step"].shape == ()
        assert float(sd_h["state"][k]["step"]) == float(sd_t["state"][k]["step"]) == 3.0
    ps_b = [torch.nn.Parameter(p.detach().clone()) for p in ps_t]
    back = torch.optim.Adam(ps_b)
    back.load_state_dict(sd_h)                  # torch.optim.Adam accepts ours and can keep stepping
    for a, b in zip(ps_t[:2], ps_b[:2]):
        a.grad = torch.ones_like(a)
        b.grad = torch.ones_like(b)
    adam.step()
    back.step()
    for a, b in zip(ps_t, ps_b):
        assert torch.equal(a, b)
    bad = copy.deepcopy(adam.state_dict())
    bad["param_groups"][0]["weight_decay"] = 0.1
    with pytest.raises(ValueError):
        hip.load_state_dict(bad)


def test_text_encoder_classes_mirror_reference_state():
    """RNN_ENCODER: upstream's parameter names/shapes (encoder.py:93-104), so a DAMSM text_encoder100.pth (an nn.Embedding +
    nn.LSTM state dict) loads strictly; GRU and SBERT fail loudly at construction."""
    from xmc_gan.config import gan
    from xmc_gan.model.encoder import RNN_ENCODER, SBERT_ENCODER
    gan.reset_cfg()
    gan.cfg_from_file(os.path.join(CFG_DIR, "df_gan_damsm.yml"))
    cfg = gan.cfg
    enc = RNN_ENCODER(cfg)
    assert {k: tuple(v.shape) for k, v in enc.state_dict().items()} == X.rnn_encoder_shapes(cfg.TEXT.VOCA_SIZE, cfg.TEXT.EMBEDDING_DIM)
    assert float(enc.encoder.weight.abs().max()) <= 0.1
    ref = torch.nn.ModuleDict(dict(encoder=torch.nn.Embedding(cfg.TEXT.VOCA_SIZE, 300),
                                   rnn=torch.nn.LSTM(300, 128, 1, batch_first=True, bidirectional=True)))
    enc.load_state_dict(ref.state_dict(), strict=True)
    with pytest.raises(RuntimeError):                    # no CPU fallback
        enc.eval()(torch.ones(2, cfg.TEXT.MAX_LENGTH, dtype=torch.int64), torch.tensor([3, 4]))
    cfg.TEXT.RNN_TYPE = "GRU"                            # encoder.py:99-102: the same container with an nn.GRU
    gru = RNN_ENCODER(cfg)
    ref_g = torch.nn.ModuleDict(dict(encoder=torch.nn.Embedding(cfg.TEXT.VOCA_SIZE, 300),
                                     rnn=torch.nn.GRU(300, 128, 1, batch_first=True, bidirectional=True)))
    gru.load_state_dict(ref_g.state_dict(), strict=True)
    cfg.TEXT.RNN_TYPE = "RNN_TANH"
    with pytest.raises(NotImplementedError):             # encoder.py:103
        RNN_ENCODER(cfg)
    with pytest.raises(ImportError):
        SBERT_ENCODER(cfg)
    gan.reset_cfg()


def test_datasets_mirror_reference_item_layout(tmp_path):
    """WordTextDataset / SentTextDataset on a miniature COCO-style tree (file layout of dataset.py:67-74,84-91,118-124):
    item = (img [3,S,S] in [-1,1], [(caption, length)], key), caption index idx*CAPTIONS_PER_IMAGE+1 (dataset.py:49-51), zero
    padding / truncation to MAX_LENGTH (104-111), and the default collate gives the batch tuple train() unpacks (train_gan.py:176-181)."""
    import pickle
    from PIL import Image
    from xmc_gan.config import gan
    from xmc_gan import dataset as D
    gan.reset_cfg()
    gan.cfg_from_file(os.path.join(CFG_DIR, "df_gan_damsm.yml"))
    cfg = gan.cfg
    cfg.IMG.SIZE, cfg.TEXT.MAX_LENGTH = 32, 6
    root = tmp_path / "coco"
    (root / "images").mkdir(parents=True)
    rng = np.random.RandomState(0)
    keys = [f"img{i}" for i in range(3)]
    for k, (w, h) in zip(keys, ((80, 50), (40, 64), (38, 38))):
        Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(root / "images" / f"{k}.jpg")
    for mode in ("train", "test"):
        (root / mode).mkdir()
        with open(root / mode / "filenames.pickle", "wb") as f:
            pickle.dump(keys, f)
    caps = [list(range(1 + j, 1 + j + (3 + j % 7))) for j in range(3 * 5)]          # lengths 3..9, ids >= 1
    i2w = {i: f"w{i}" for i in range(40)}
    with open(root / "captions.pickle", "wb") as f:
        pickle.dump([caps, caps[::-1], i2w, {v: k for k, v in i2w.items()}], f)
    sents = [f"a photo of thing number {j}" for j in range(15)]
    with open(root / "bert_captions.pickle", "wb") as f:
        pickle.dump([sents, sents[::-1]], f)

    torch.manual_seed(0)
    ds = D.WordTextDataset(str(root), "train", D.train_transform(32), cfg)
    assert len(ds) == 3 and ds.voca_size == 40
    for idx in range(3):
        img, texts, key = ds[idx]
        assert img.shape == (3, 32, 32) and img.dtype == torch.float32 and -1.0 <= float(img.min()) and float(img.max()) <= 1.0
        assert key == keys[idx] and len(texts) == 1
        cap, n = texts[0]
        src = caps[idx * 5 + 1]
        assert cap.dtype == np.int64 and cap.shape == (6,) and n == min(len(src), 6)
        assert list(cap[:n]) == src[:n] and (cap[n:] == 0).all()
    assert D.index_to_sent(i2w, [ds[0][1][0][0]]) == [" ".join(f"w{t}" for t in caps[1][:6])]
    imgs, texts_lst, ks = next(iter(torch.utils.data.DataLoader(ds, batch_size=2, drop_last=True, shuffle=False)))
    caps_b, lens_b = texts_lst[0]
    assert imgs.shape == (2, 3, 32, 32) and caps_b.shape == (2, 6) and caps_b.dtype == torch.int64 and lens_b.shape == (2,)
    assert list(ks) == keys[:2]
    # test split: exact resize, reversed caption table
    dt = D.WordTextDataset(str(root), "test", D.test_transform(32), cfg)
    img, texts, _ = dt[2]
    assert img.shape == (3, 32, 32) and list(texts[0][0][:texts[0][1]]) == caps[::-1][2 * 5 + 1][:6]
    # transforms: shorter side -> 76/64 of the crop, as train_gan.py:443-447
    im = D.Resize(38)(Image.new("RGB", (80, 50)))
    assert im.size == (60, 38)
    assert D.Resize((32, 32))(Image.new("RGB", (80, 50))).size == (32, 32)
    t = D.to_normalized_tensor(Image.fromarray(np.array([[[0, 255, 51]]], dtype=np.uint8)))
    assert torch.allclose(t.flatten(), torch.tensor([-1.0, 1.0, -0.6]), atol=1e-6)
    sd = D.SentTextDataset(str(root), "train", D.test_transform(32), cfg)
    assert sd[1][1] == [(sents[6], 6)]
    with pytest.raises(NotImplementedError):
        D.WordTextDataset(str(tmp_path / "nowhere"), "train", None, cfg)
    gan.reset_cfg()


def test_word_attention_generator_state_dict_and_registry():
    """concept_gan.OutNetG: state_dict keys/shapes equal the reference's (pinned through
    fwd_wordg*.npz's key table by test_oracle_golden) for every image size, with and without normalisation; the same for the repaired InNetG."""
    from xmc_gan.config import gan
    import xmc_gan.train_gan as tg
    gan.reset_cfg()
    gan.cfg_from_file(os.path.join(CFG_DIR, "df_gan_damsm_nomagp.yml"))
    cfg = gan.cfg
    cfg.TRAIN.NCH = 8
    # CONCEPT_INATTN_GEN: concept_gan.InNetG with the two documented repairs (fwd_wordin*.npz's key table comes from the reference's
    # own constructor plus those patches, oracle/make_golden.py repaired_word_in_netg)
    for name in ("CONCEPT_OUTATTN_GEN", "CONCEPT_INATTN_GEN"):
        cfg.GEN.ENCODER_NAME = name
        for size in (64, 128, 256):
            for norm in (True, False):
                cfg.IMG.SIZE, cfg.GEN.NORMALIZE = size, norm
                h = X.Hyper.from_cfg(cfg)
                netG = tg._GEN_ARCH[cfg.GEN.ENCODER_NAME](cfg)
                sd = netG.state_dict()
                assert {k: tuple(v.shape) for k, v in sd.items()} == X.gen_shapes(h)
                assert all(v.dtype == torch.int64 for k, v in sd.items() if k.endswith("num_batches_tracked"))
                netG.load_state_dict(X.synth_params(X.gen_shapes(h), 1), strict=True)
    gan.reset_cfg()


def test_product_never_touches_the_oracle_or_a_cpu_fallback():
    """The oracle is test infrastructure: nothing under xmc_gan/ or xmc-gan_amd/ may import it (only tests/, smoke() and
    bench.py's cpu_baseline leg do), and nothing there may read the reference tree."""
    bad = []
    for top in ("xmc_gan", "xmc-gan_amd"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"^\s*(import|from)\s+(xmc_ref|ref_harness|oracle)\b", txt, re.M) or "/root/reference" in txt:
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert "xmc_ref" in entry and "xmc_ref" in bench                      # the two sanctioned users
    assert "/root/reference" not in bench                                  # nothing at run time reads the reference


def test_cli_flags_match_reference():
    import xmc_gan.train_gan as tg
    a = tg.parse_args([])
    assert (a.cfg, a.gpu_id, a.seed, a.resume_epoch, a.log_type, a.bs, a.imsize) == \
        ('xmc_gan/cfg/df_gan_sbert_seperate.yml', 0, 100, 0, 'tb', -1, -1)
    # the reference's live registry (train_gan.py:42-49) plus the two word-attention names it keeps commented out
    assert set(tg._GEN_ARCH) == {"DF_GEN", "CONCEPT_IN_DF_GEN", "CONCEPT_OUT_DF_GEN", "CONCEPT_OUTATTN_GEN", "CONCEPT_INATTN_GEN"}
    assert set(tg._DISC_ARCH) == {"DF_DISC", "CONCEPT_NETD"}


def test_bench_gpus_n_without_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` outside torchrun must start two ranks itself (child process) instead of silently timing one:
    without a GPU each rank stops at the "needs an MI355X" check, which proves both were started with WORLD_SIZE=2."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU rehearsal")
    assert r.returncode != 0
    # (torchrun tears the other rank down as soon as the first one has failed, so one message is guaranteed, two are not)
    assert "needs an MI355X" in r.stderr and "torch.distributed" in r.stderr, r.stderr[-2000:]
    # and a mismatching launcher is refused instead of being ignored
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env2, capture_output=True, text=True,
                        timeout=300)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in (r2.stderr + r2.stdout)


def test_image_grid_and_scalar_log(tmp_path):
    """the logging tail's host helpers (reference train_gan.py:160,299-326 via torchvision.utils.save_image / tensorboard): grid
    geometry, per-image min-max scaling, PNG round trip, JSON-lines scalars"""
    from PIL import Image
    from xmc_gan.utils.visual import ScalarLog, make_grid, save_image, to_uint8_hwc
    rng = np.random.RandomState(0)
    x = rng.randn(10, 3, 6, 5).astype(np.float32) * 3
    g = make_grid(x)                                    # 8 per row -> 2 rows; padding 2
    assert g.shape == (3, 2 * (6 + 2) + 2, 8 * (5 + 2) + 2)
    tile = g[:, 2:8, 2:7]                               # first image, scaled to [0, 1] on its own
    assert abs(tile.min()) < 1e-6 and abs(tile.max() - 1.0) < 1e-6
    np.testing.assert_allclose(tile, (x[0] - x[0].min()) / (x[0].max() - x[0].min()), atol=1e-6)
    assert np.all(g[:, :2, :] == 0)                      # padding stays at pad_value
    np.testing.assert_allclose(g[:, 10:16, 9:14], (x[9] - x[9].min()) / (x[9].max() - x[9].min()), atol=1e-6)   # image 9: row 1, cell 1
    assert np.all(g[:, 10:16, 16:] == 0)                 # empty cells of the last row
    save_image(torch.from_numpy(x), tmp_path / "g.png")
    im = np.asarray(Image.open(tmp_path / "g.png"))
    assert im.shape == (g.shape[1], g.shape[2], 3) and im.dtype == np.uint8
    np.testing.assert_allclose(im.transpose(2, 0, 1) / 255.0, g, atol=1 / 255.0 + 1e-6)
    # the training loop's form: the batch is copied when the call is made (the caller may overwrite its tensor straight away -- a graph's
    # static output), the file is written by the writer thread and is there after flush_saves(); same bytes as the inline writer's
    from xmc_gan.utils.visual import flush_saves, save_image_async
    t = torch.from_numpy(x.copy())
    save_image_async(t, tmp_path / "g_async.png")
    t.zero_()
    save_image_async(np.zeros((1, 5, 4, 4), dtype=np.float32), tmp_path / "bad.png")        # 5 channels: no such image -- reported at the flush
    with pytest.raises(Exception):
        flush_saves()
    assert (tmp_path / "g_async.png").read_bytes() == (tmp_path / "g.png").read_bytes()
    flush_saves()                                                                             # nothing pending, nothing to report
    assert to_uint8_hwc(np.zeros((3, 4, 4), np.float32)).tolist() == np.full((4, 4, 3), 127, np.uint8).tolist()   # truncation, as upstream
    log = ScalarLog(str(tmp_path), "tb")
    log.add_scalar("Loss_D", 1.5, 3)
    log.add_scalar("FID", 42.0, 3)
    log.close()
    rows = [json.loads(line) for line in open(tmp_path / "scalars.jsonl")]
    assert rows == [{"tag": "Loss_D", "value": 1.5, "step": 3}, {"tag": "FID", "value": 42.0, "step": 3}]


def test_integration_doc_stub_mirrors_the_descriptor():
    """INTEGRATION.md shows a maintainer the ctypes mirror of XmcConvDesc: the struct in that code block must be the real one, field for
    field (the library reads every field; a stub that stops short hands it stack garbage as option pointers)."""
    import ctypes as C
    import re
    from xmc_gan_amd import lib as L
    txt = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    m = re.search(r"(class XmcConvDesc\(C\.Structure\):.*?)\nassert C\.sizeof\(XmcConvDesc\) == (\d+)", txt, re.S)
    assert m, "the XmcConvDesc stub is gone from INTEGRATION.md"
    ns = {"C": C}
    exec(m.group(1), ns)
    doc = ns["XmcConvDesc"]
    assert int(m.group(2)) == C.sizeof(doc) == C.sizeof(L.ConvDesc)
    assert [(n, C.sizeof(t)) for n, t in doc._fields_] == [(n, C.sizeof(t)) for n, t in L.ConvDesc._fields_]
    assert f"xmc_abi_version() == {L.ABI_VERSION}" in txt


def test_public_header_is_plain_c(tmp_path):
    """include/xmc_gan_hip.h is the drop-in boundary: it must compile as C99 (and as C++) on its own, and a C translation unit must see the
    struct size the Python binding mirrors."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    from xmc_gan_amd import lib as L
    hdr = os.path.join(ROOT, "include", "xmc_gan_hip.h")
    subprocess.run(["gcc", "-x", "c", "-std=c99", "-fsyntax-only", "-Wall", "-Werror", hdr], check=True)
    subprocess.run(["g++", "-x", "c++", "-fsyntax-only", hdr], check=True)
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "xmc_gan_hip.h"\nint main(void) { printf("%zu %d\\n", sizeof(XmcConvDesc), XMC_ABI_VERSION); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    size, abi = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(size) == ctypes.sizeof(L.ConvDesc) and int(abi) == L.ABI_VERSION
