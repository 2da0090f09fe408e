"""timing of the contrastive head (not a test): python tests/diag/contrastive_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from xmc_gan_amd import ops
for n, D in ((256, 256), (2048, 256), (2048, 512)):
    a = torch.randn(n, D, device="cuda", requires_grad=True)
    b = torch.randn(n, D, device="cuda", requires_grad=True)
    for _ in range(3):
        l = ops.contrastive(a, b); l.backward()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    reps = 20
    e[0].record()
    for _ in range(reps):
        l = ops.contrastive(a, b)
    e[1].record()
    for _ in range(reps):
        l = ops.contrastive(a, b); l.backward()
    e[2].record()
    torch.cuda.synchronize()
    f = e[0].elapsed_time(e[1]) / reps
    fb = e[1].elapsed_time(e[2]) / reps
    print(f"n={n} D={D}: forward {f*1e3:.0f} us, forward+backward {fb*1e3:.0f} us (eager, includes launch gaps)")
