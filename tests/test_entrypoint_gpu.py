"""The cfg-driven entry point (``xmc_gan/train_gan.py``, reference train_gan.py:398-498) end to end on the GPU: synthetic
batches through the real RNN_ENCODER, the real-data path (WordTextDataset + DataLoader + RNN_ENCODER) on a miniature
COCO-style tree, and checkpoint save (epoch > 50, train_gan.py:328-334) followed by ``--resume_epoch``."""
import math
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import CFG_DIR


def _mini_yml(tmp_path, **subst):
    """df_gan_damsm.yml shrunk for a test run (thin network, tiny vocabulary, no pretrained encoder file)."""
    txt = open(os.path.join(CFG_DIR, "df_gan_damsm.yml")).read()
    rep = {"NCH: 32": "NCH: 8", "VOCA_SIZE: 27297": "VOCA_SIZE: 40", "BATCH_SIZE: 88": "BATCH_SIZE: 4", "LOG_INTERVAL: 200": "LOG_INTERVAL: 2",
           "NUM_WORKERS: 8": "NUM_WORKERS: 0", "ENCODER_DIR: data/DAMSMencoders/coco/text_encoder100.pth": "ENCODER_DIR: ''",
           "MAX_LENGTH: 20": "MAX_LENGTH: 8", "MAGP: true": "MAGP: false"}
    rep.update(subst)
    for a, b in rep.items():
        assert a in txt, a
        txt = txt.replace(a, b)
    path = tmp_path / "mini.yml"
    path.write_text(txt)
    return str(path)


def _mini_coco(root, n_img=8):
    from PIL import Image
    rng = np.random.RandomState(1)
    (root / "images").mkdir(parents=True)
    keys = [f"k{i:03d}" for i in range(n_img)]
    for k in keys:
        Image.fromarray(rng.randint(0, 256, (90, 100, 3), dtype=np.uint8)).save(root / "images" / f"{k}.jpg")
    for mode in ("train", "test"):
        (root / mode).mkdir()
        with open(root / mode / "filenames.pickle", "wb") as f:
            pickle.dump(keys, f)
    caps = [list(rng.randint(1, 40, size=rng.randint(2, 12))) for _ in range(n_img * 5)]
    i2w = {i: f"w{i}" for i in range(40)}
    with open(root / "captions.pickle", "wb") as f:
        pickle.dump([caps, caps, i2w, {v: k for k, v in i2w.items()}], f)
    return str(root)


def _finite(last):
    assert {"errD", "errG"} <= set(last)
    for k, v in last.items():
        if torch.is_tensor(v) and v.numel() == 1:
            assert math.isfinite(float(v)), k


def test_synthetic_run_uses_the_hip_rnn_encoder(tmp_path):
    import xmc_gan.train_gan as tg
    last = tg.main(["--cfg", _mini_yml(tmp_path), "--synthetic", "3", "--max_epoch", "1", "--precision", "bf16",
                    "--output_dir", str(tmp_path / "run")])
    _finite(last)


@pytest.mark.parametrize("gen", ["CONCEPT_OUTATTN_GEN", "CONCEPT_INATTN_GEN", "CONCEPT_IN_DF_GEN", "CONCEPT_OUT_DF_GEN"])
@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_word_attention_generators_from_the_entry_point(tmp_path, gen, precision):
    """The word-attention generators consume what the reference's loop hands every generator -- the RNN_ENCODER's per-word embeddings
    and padding mask (train_gan.py:160-170, 197) -- selected by GEN.ENCODER_NAME like any other (the names upstream left commented out,
    train_gan.py:31,44; InNetG is the documented repair): captions of different lengths, 64 px, both 16-bit modes, two iterations.  The two
    attention-modulation generators ride along: their hoisted sentence products (one launch for all stages) under the real loop, graph capture
    and the half mode's loss scale."""
    import xmc_gan.train_gan as tg
    yml = _mini_yml(tmp_path, **{"ENCODER_NAME: DF_GEN": f"ENCODER_NAME: {gen}"})
    last = tg.main(["--cfg", yml, "--synthetic", "2", "--max_epoch", "1", "--precision", precision, "--output_dir", str(tmp_path / "run")])
    _finite(last)


def test_real_data_path_and_resume(tmp_path):
    import xmc_gan.train_gan as tg
    data, run = _mini_coco(tmp_path / "coco"), str(tmp_path / "run")
    yml = _mini_yml(tmp_path, **{"MAX_EPOCH: 121": "MAX_EPOCH: 52"})
    last = tg.main(["--cfg", yml, "--data_dir", data, "--output_dir", run, "--precision", "fp32", "--seed", "3"])
    _finite(last)
    saved = sorted(os.listdir(os.path.join(run, "model")))
    assert saved == ["netD_051.pth", "netD_052.pth", "netG_051.pth", "netG_052.pth", "optimizerD.pth", "optimizerG.pth"]
    sd = torch.load(os.path.join(run, "model", "optimizerD.pth"), map_location="cpu")
    assert all(float(st["step"]) == 52 * 2 for st in sd["state"].values())           # 2 batches of 4 per epoch, 52 epochs
    # resume: loads the four files and trains epoch 53 only
    yml2 = _mini_yml(tmp_path, **{"MAX_EPOCH: 121": "MAX_EPOCH: 53"})
    last2 = tg.main(["--cfg", yml2, "--data_dir", data, "--output_dir", run, "--precision", "fp32", "--seed", "3",
                     "--resume_epoch", "52"])
    _finite(last2)
    sd2 = torch.load(os.path.join(run, "model", "optimizerD.pth"), map_location="cpu")
    assert all(float(st["step"]) == 53 * 2 for st in sd2["state"].values())
    assert "netG_053.pth" in os.listdir(os.path.join(run, "model"))
    # the logging / evaluation tail (reference train_gan.py:146-160, 297-326, 338-395)
    img = os.path.join(run, "img")
    files = set(os.listdir(img))
    assert {"sents.txt", "imgs.png", "fake_samples_epoch_001.png", "fake_samples_epoch_053.png", "test", "org"} <= files, files
    assert any(f.startswith("fake_samples_") and "epoch" not in f for f in files)          # every LOG_INTERVAL steps
    assert len(open(os.path.join(img, "sents.txt")).read().splitlines()) == 4             # one caption per sample of the first batch
    assert len(os.listdir(os.path.join(img, "test"))) == len(os.listdir(os.path.join(img, "org"))) == 8    # the 8 test images
    from PIL import Image
    assert Image.open(os.path.join(img, "test", "k000.png")).size == (64, 64)
    rows = [__import__("json").loads(line) for line in open(os.path.join(run, "log", "scalars.jsonl"))]
    tags = {r["tag"] for r in rows}
    assert {"epoch", "Loss_D", "Loss_G", "errD_real", "errD_fake", "errD_mismatch", "ds_loss", "gs_loss", "disc_loss"} <= tags, tags
    assert max(r["step"] for r in rows) == 53


@pytest.mark.parametrize("case", ["headline losses", "MA-GP", "N_CRITIC = 2"])
def test_entry_point_graph_replay_equals_eager_launches(tmp_path, case):
    """`train()` behind `main()` replays the iteration as hipGraphs by default (two eager warm-ups, one capture per N_CRITIC phase, then
    replays with the loader's batches copied into the static inputs).  Six iterations through `main()` with `--graph 1` and with
    `--graph 0` from the same seed, fp32 mode: same final weights (to the f32 atomics order of the weight-gradient kernels: 1e-5 per
    tensor, see below) and the same last losses -- for the headline loss set, with the gradient penalty, and with two critic steps per generator step
    (two captured phases)."""
    import xmc_gan.train_gan as tg
    subst = {"MAGP: true": "MAGP: true" if case == "MA-GP" else "MAGP: false"}
    if case == "N_CRITIC = 2":
        subst["N_CRITIC: 1"] = "N_CRITIC: 2"
    yml = _mini_yml(tmp_path, **subst)
    res = {}
    for graph in (1, 0):
        last = tg.main(["--cfg", yml, "--synthetic", "6", "--max_epoch", "1", "--precision", "fp32", "--seed", "11", "--graph", str(graph),
                        "--output_dir", str(tmp_path / f"run{graph}")])
        netG, netD = tg.main.last_models
        res[graph] = (last, {"G." + k: v.detach().float().cpu().clone() for k, v in netG.state_dict().items()} |
                      {"D." + k: v.detach().float().cpu().clone() for k, v in netD.state_dict().items()})
    assert res[1][0].get("hipgraph") is True and "hipgraph" not in res[0][0]
    # Adam with beta1 = 0 moves an element by +-lr whatever the size of its gradient: an element whose gradient is zero to within the
    # summation-order noise can step the other way in one of the two runs (seen once in ~15 runs: ONE element of an 8 192-element tensor,
    # 1.4e-4 of the tensor's norm on its own).  So: the tensor without its 1-in-10 000 most different elements (at least one) to 1e-5, and
    # those elements within what six such steps can differ by (12 x the larger learning rate).
    worst, flipped = 0.0, 0
    for k, a in res[1][1].items():
        b = res[0][1][k]
        d = (a - b).abs().flatten().double()
        top = d.topk(max(1, d.numel() // 10000)).values
        e = ((d.square().sum() - top.square().sum()).clamp_min(0).sqrt() / b.norm().double().clamp_min(1e-12)).item()
        worst = max(worst, e)
        assert e <= 1e-5, (k, e)
        assert top.max().item() <= 12 * 4e-4, (k, top.max().item())
        flipped += int((top > 1e-4).sum())
    for k in ("errD", "errG", "errD_real", "errD_fake"):
        a, b = float(res[1][0][k]), float(res[0][0][k])
        assert abs(a - b) <= (2e-4 if flipped else 1e-5) * abs(b) + 1e-6, (k, a, b)      # (one element stepping the other way: 1e-4 of a layer)
    print(f"\n[entry point, {case}] graph replay vs eager launches after 6 iterations: worst parameter tensor {worst:.1e}"
          f" ({flipped} elements stepped the other way)")
