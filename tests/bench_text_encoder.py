"""Micro-benchmark of the frozen text front end (not a test): RNN_ENCODER forward at the headline batch, HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xmc_gan"))
from xmc_gan.config import gan  # noqa: E402
from xmc_gan.model.encoder import RNN_ENCODER  # noqa: E402

gan.reset_cfg()
gan.cfg_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xmc_gan", "cfg", "df_gan_damsm.yml"))
cfg = gan.cfg
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
enc = RNN_ENCODER(cfg).cuda().eval()
g = torch.Generator().manual_seed(0)
lens = torch.randint(5, cfg.TEXT.MAX_LENGTH + 1, (B,), generator=g)
caps = torch.randint(1, cfg.TEXT.VOCA_SIZE, (B, cfg.TEXT.MAX_LENGTH), generator=g) * (torch.arange(cfg.TEXT.MAX_LENGTH)[None] < lens[:, None])
caps_d, lens_d = caps.cuda(), lens.cuda()
for _ in range(5):
    enc(caps_d, lens_d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 50
e0.record()
for _ in range(N):
    enc(caps_d, lens_d)
e1.record()
torch.cuda.synchronize()
print(f"RNN_ENCODER forward B={B} T={cfg.TEXT.MAX_LENGTH}: {e0.elapsed_time(e1) / N * 1e3:.1f} us per call (device-resident captions)")
