"""timing of the contrastive head (not a test): python tests/diag/contrastive_time.py
Device time of xmc_contrastive_fwd / _bwd as they run inside the replayed iteration: ten evaluations captured in one hipGraph, replayed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from xmc_gan_amd import ops
from xmc_gan_amd import lib as L
p = lambda t: None if t is None else t.data_ptr()
for n, D in ((256, 256), (2048, 256), (2048, 512)):
    a, b = torch.randn(n, D, device="cuda"), torch.randn(n, D, device="cuda")
    ws = torch.empty(L.load().xmc_contrastive_ws_bytes(n, D), dtype=torch.uint8, device="cuda")
    loss, g = torch.empty(1, device="cuda"), torch.ones(1, device="cuda")
    da, db = torch.empty_like(a), torch.empty_like(b)

    def run(bwd):
        st = torch.cuda.current_stream().cuda_stream
        L.call("xmc_contrastive_fwd", p(a), p(b), None, None, n, D, p(loss), p(ws), st)
        if bwd:
            L.call("xmc_contrastive_bwd", p(a), p(b), None, None, n, D, p(g), p(ws), p(da), p(db), st)
    run(True)
    torch.cuda.synchronize()
    res = {}
    for what in (False, True):
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            gr.capture_begin()
            for _ in range(10):
                run(what)
            gr.capture_end()
        torch.cuda.synchronize()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        res[what] = e0.elapsed_time(e1) / 100 * 1e3
    print(f"n={n} D={D}: forward {res[False]:.1f} us, forward + backward {res[True]:.1f} us (device time, hipGraph replay)")
