"""Which Python call sites launch the small ATen kernels (fills, copies, adds) of one steady-state eager G+D iteration of the bench
configuration (diagnostic: prints a table).  python tests/diag/fill_sites.py [batch] [yml] [img size]"""
import os, sys, traceback, collections
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import importlib
ops = importlib.import_module("xmc-gan_amd.ops")
from parity_util import setup_cfg, build_product, X, DEV
import xmc_gan.train_gan as tg

ops.set_precision("bf16")
cfg, h = setup_cfg(sys.argv[2] if len(sys.argv) > 2 else "df_gan_damsm_nomagp.yml", **{"IMG.SIZE": int(sys.argv[3]) if len(sys.argv) > 3 else 256})
PG, PD = X.synth_params(X.gen_shapes(h), 3), X.synth_params(X.netd_shapes(h), 4)
models = build_product(h, PG, PD)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
b = {k: v.to(DEV) for k, v in X.synth_batch(h, B, seed=9, words_len=cfg.TEXT.MAX_LENGTH).items()}
st = {}
for _ in range(3):
    tg.gan_iteration(*models, b["imgs"], b["sent_embs"], b["words_embs"], b["mask"], b["noise"], st)
torch.cuda.synchronize()
sites = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(t in name for t in ("fill", "zero", "ones", "copy_", "clone", "add", "mul", "cat", "neg", "_to_copy", "full")) \
                and "like" not in os.environ.get("FILL_SITES_ONLY", "like") or any(t in name for t in ("fill", "zero", "ones", "full")):
            fr = [f for f in traceback.extract_stack() if "/root/repo" in f.filename or "repo/" in f.filename]
            fr = [f for f in fr if "fill_sites" not in f.filename]
            where = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-4:])
            shp = [tuple(a.shape) for a in args if torch.is_tensor(a)][:2]
            sites[(name, where, str(shp))] += 1
        return func(*args, **(kwargs or {}))


with torch.autograd.set_multithreading_enabled(False), Spy():       # backward nodes on this thread, so the mode sees them
    tg.gan_iteration(*models, b["imgs"], b["sent_embs"], b["words_embs"], b["mask"], b["noise"], st)
torch.cuda.synchronize()
for (name, where, shp), n in sites.most_common(70):
    print(f"{n:4d}  {name:34s} {shp:40s} {where}")
