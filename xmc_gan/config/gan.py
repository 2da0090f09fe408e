"""Global experiment configuration (same schema, defaults and merge rules as the reference's
``xmc_gan/config/gan.py:7-131``; the yml presets under ``xmc_gan/cfg`` load unchanged).

Differences that do not change results: the attribute-dict is implemented here (the reference
imports ``easydict``) and yml files are read with ``yaml.safe_load`` (the reference's bare
``yaml.load(f)`` no longer exists in PyYAML 6).
"""
import numpy as np


class AttrDict(dict):
    """dict with attribute access; nested dicts are converted on assignment."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


edict = AttrDict


def _defaults():
    return AttrDict({
        "CONFIG_NAME": "", "DATASET_NAME": "coco",
        "TRAIN": {
            "FLAG": True, "MAX_EPOCH": 1000, "BATCH_SIZE": 256, "NUM_WORKERS": 8, "LOG_INTERVAL": 1,
            "SAVE_INTERVAL": 1, "N_CRITIC": 1, "HE_INIT": False, "NEF": 128, "NCH": 32, "NOISE_DIM": 128,
            "RMIS_LOSS": False, "MAGP": False,
            "ENCODER_LOSS": {"B_GLOBAL": False, "SENT": False, "WORD": False, "DISC": False, "VGG": False},
            "SMOOTH": {"MISMATCH": 1.0, "GLOBAL": 0.5, "SENT": 1.0, "DISC": 1.0},
            "OPT": {"G_LR": 0.0001, "G_BETA1": 0.5, "G_BETA2": 0.999, "D_LR": 0.0004, "D_BETA1": 0.5, "D_BETA2": 0.999},
        },
        "GEN": {"ENCODER_NAME": "", "NORMALIZE": True},
        "DISC": {"ENCODER_NAME": "", "ENCODER_DIR": "", "SPEC_NORM": True, "UNCOND": True, "COND": True,
                 "SENT_MATCH": False, "IMG_MATCH": False, "SEPERATE": False},
        "IMG": {"SIZE": 64},
        "TEXT": {"TYPE": "WORD", "CAPTIONS_PER_IMAGE": 5, "MAX_LENGTH": 20, "VOCA_SIZE": 27297,
                 "ENCODER_NAME": "RNN", "ENCODER_DIR": "", "EMBEDDING_DIM": 256, "NUM_LAYERS": 1, "RNN_TYPE": "LSTM",
                 "FIX_BERT": True, "BERT_NORM": False, "POOLING_MODE": "MEAN", "SENT_FT": False, "WORD_FT": False,
                 "JOINT_FT": False},
    })


__C = _defaults()
cfg = __C


def reset_cfg():
    """Restore the defaults in place (every module holds a reference to the same object)."""
    for k in list(__C.keys()):
        del __C[k]
    for k, v in _defaults().items():
        __C[k] = v
    return __C


def _merge_a_into_b(a, b):
    """Overlay ``a`` on ``b``: unknown key -> KeyError, type mismatch -> ValueError (config/gan.py:92-122)."""
    if type(a) is not AttrDict:
        return
    for k, v in a.items():
        if k not in b:
            raise KeyError('{} is not a valid config key'.format(k))
        old_type = type(b[k])
        if old_type is not type(v):
            if isinstance(b[k], np.ndarray):
                v = np.array(v, dtype=b[k].dtype)
            else:
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(b[k]), type(v), k))
        if type(v) is AttrDict:
            try:
                _merge_a_into_b(a[k], b[k])
            except Exception:
                print('Error under config key: {}'.format(k))
                raise
        else:
            b[k] = v


def cfg_from_file(filename):
    """Load a yml preset and merge it into the global options."""
    import yaml
    with open(filename, 'r') as f:
        yaml_cfg = AttrDict(yaml.safe_load(f))
    _merge_a_into_b(yaml_cfg, __C)
