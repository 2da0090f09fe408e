"""Adam on the multi-tensor HIP kernel (torch.optim.Adam semantics of train_gan.py:483-484: eps 1e-8, no weight
decay, parameters whose ``.grad`` is None are skipped and keep their step count)."""
import ctypes as C

import torch

from . import lib as L
from . import ops


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._tables = {}
        self._chunk = None

    def _state_for(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.zeros(1, dtype=torch.int32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
        return st

    def _table(self, ps, device):
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
        hit = self._tables.get(key)
        if hit is not None:
            return hit
        if self._chunk is None:
            self._chunk = L.load().xmc_adam_chunk_elems()
        ents = (L.AdamEntry * len(ps))()
        chunks = []
        for i, p in enumerate(ps):
            st = self._state_for(p)
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous() or not p.is_contiguous():
                raise RuntimeError("HipAdam needs contiguous f32 parameters and gradients")
            ents[i].param, ents[i].grad = p.data_ptr(), g.data_ptr()
            ents[i].m, ents[i].v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            ents[i].step, ents[i].n = st["step"].data_ptr(), p.numel()
            chunks += [(i, c) for c in range((p.numel() + self._chunk - 1) // self._chunk)]
        tab = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8).to(device)
        ch = torch.tensor(chunks, dtype=torch.int32).to(device)
        if len(self._tables) > 16:
            self._tables.clear()
        self._tables[key] = (tab, ch, len(ps), len(chunks))
        return self._tables[key]

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            if not ps[0].is_cuda:
                raise RuntimeError("HipAdam runs on the GPU only (no CPU fallback)")
            tab, ch, nt, nc = self._table(ps, ps[0].device)
            b1, b2 = group["betas"]
            L.call("xmc_adam_step", C.c_void_p(tab.data_ptr()), nt, C.c_void_p(ch.data_ptr()), nc,
                   float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
        ops.bump_weights_epoch()
