"""Helpers shared by the golden-vector tests (test infrastructure)."""
import os

import numpy as np
import yaml

import xmc_ref as X

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CFG_DIR = os.path.join(ROOT, "xmc_gan", "cfg")


def _deep_update(dst, src):
    for k, v in src.items():
        if isinstance(v, dict):
            _deep_update(dst.setdefault(k, {}), v)
        else:
            dst[k] = v


DEFAULTS = dict(  # the reference defaults the yml files rely on (config/gan.py:7-90)
    TRAIN=dict(N_CRITIC=1, NEF=128, NCH=32, NOISE_DIM=128, RMIS_LOSS=False, MAGP=False,
               ENCODER_LOSS=dict(B_GLOBAL=False, SENT=False, WORD=False, DISC=False, VGG=False),
               SMOOTH=dict(MISMATCH=1.0, GLOBAL=0.5, SENT=1.0, DISC=1.0),
               OPT=dict(G_LR=1e-4, G_BETA1=0.5, G_BETA2=0.999, D_LR=4e-4, D_BETA1=0.5, D_BETA2=0.999)),
    GEN=dict(ENCODER_NAME="", NORMALIZE=True),
    DISC=dict(ENCODER_NAME="", SPEC_NORM=True, SENT_MATCH=False, IMG_MATCH=False, SEPERATE=False),
    IMG=dict(SIZE=64), TEXT=dict(EMBEDDING_DIM=256, MAX_LENGTH=20))


def hyper_for(fix):
    """Hyper for a fixture: its yml (our copy under xmc_gan/cfg) + the recorded overrides."""
    import copy
    cfg = copy.deepcopy(DEFAULTS)
    with open(os.path.join(CFG_DIR, str(fix["yml"]))) as f:
        _deep_update(cfg, yaml.safe_load(f))
    for ov in fix["over"]:
        k, v = str(ov).split("=")
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        if v in ("True", "False"):
            node[parts[-1]] = v == "True"
        else:
            try:
                node[parts[-1]] = int(v)
            except ValueError:
                node[parts[-1]] = v                  # a name, e.g. GEN.ENCODER_NAME=CONCEPT_OUTATTN_GEN
    return X.Hyper.from_cfg(cfg), cfg


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def fixtures(prefix):
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def stats(t):
    f = t.detach().double().flatten()
    n = f.numel()
    return np.array([f.sum().item(), f.abs().sum().item(), f[0].item(), f[n // 2].item(), f[-1].item()])
