"""Import the *reference* XMC-GAN code (read-only mount at /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  This module exists solely so that
``oracle/make_golden.py`` can run the reference's own hot-path code in the build
container and record input/output vectors under ``tests/golden/``.  It is never
imported by the product (``xmc_gan/``, ``xmc-gan_amd/``) and cannot run on the GPU
box (the reference does not travel there).

What it does (recipe from SURVEY.md section 8c):
  * puts empty stand-in modules for the third-party packages the reference imports
    but this image lacks (torchvision, wandb, tensorboard, pytorch_fid,
    sentence_transformers, easydict) into ``sys.modules``;
  * makes ``.cuda()`` the identity so the reference's bare ``.cuda()`` calls stay on CPU;
  * exposes ``load_cfg(yml)`` which merges a yml into the reference's global cfg with
    ``yaml.safe_load`` (the reference's own ``cfg_from_file`` uses the removed
    ``yaml.load(f)`` call signature, config/gan.py:129).
"""
import copy
import os
import sys
import types

REF_ROOT = os.environ.get("XMC_REFERENCE_ROOT", "/root/reference")


class _EasyDict(dict):
    """Attribute-style dict with recursive conversion (stand-in for easydict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)

    __setitem__ = __setattr__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_installed = False


def install():
    """Install stand-ins and make the reference importable. Idempotent."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True
    import torch

    tv = _stub("torchvision")
    tv.utils = _stub("torchvision.utils", save_image=lambda *a, **k: None)
    tv.transforms = _stub("torchvision.transforms")
    _stub("wandb", watch=lambda *a, **k: None, log=lambda *a, **k: None,
          init=lambda *a, **k: None)
    _stub("tensorboard")

    class _Writer:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

    _stub("torch.utils.tensorboard", SummaryWriter=_Writer)
    pf = _stub("pytorch_fid")
    pf.fid_score = _stub("pytorch_fid.fid_score",
                         calculate_fid_given_paths=lambda *a, **k: float("nan"))
    _stub("sentence_transformers", util=types.SimpleNamespace(),
          SentenceTransformer=object, models=types.SimpleNamespace())
    _stub("easydict", EasyDict=_EasyDict)

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _installed = True


_cfg_defaults = None


def load_cfg(yml_name, **overrides):
    """Reset the reference's global cfg to defaults, merge ``cfg/<yml_name>``, apply
    dotted-key overrides (e.g. ``**{"IMG.SIZE": 64}``). Returns the global cfg."""
    global _cfg_defaults
    install()
    import yaml
    from xmc_gan.config import gan as refcfg

    if _cfg_defaults is None:
        _cfg_defaults = copy.deepcopy(refcfg.cfg)
    # restore defaults in place (reference modules hold a reference to the same object)
    for k in list(refcfg.cfg.keys()):
        del refcfg.cfg[k]
    for k, v in copy.deepcopy(_cfg_defaults).items():
        refcfg.cfg[k] = v
    path = yml_name if os.path.isabs(yml_name) else os.path.join(REF_ROOT, "xmc_gan", "cfg", yml_name)
    with open(path) as f:
        y = _EasyDict(yaml.safe_load(f))
    refcfg._merge_a_into_b(y, refcfg.cfg)
    for dotted, v in overrides.items():
        node = refcfg.cfg
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return refcfg.cfg


def modules():
    """Return the reference modules used to make goldens."""
    install()
    import xmc_gan.train_gan as tg
    import xmc_gan.model.df_gan as df_gan
    import xmc_gan.model.df_concept_gan as df_concept_gan
    import xmc_gan.model.encoder as encoder
    import xmc_gan.model.concept_gan as concept_gan
    return types.SimpleNamespace(train_gan=tg, df_gan=df_gan, df_concept_gan=df_concept_gan, encoder=encoder,
                                 concept_gan=concept_gan)
