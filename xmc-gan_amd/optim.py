"""Adam on the multi-tensor HIP kernel (torch.optim.Adam semantics of train_gan.py:483-484: eps 1e-8, no weight
decay, parameters whose ``.grad`` is None are skipped and keep their step count)."""
import collections
import ctypes as C

import torch

from . import lib as L
from . import ops


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        # weight_decay / amsgrad are carried (at the only values the path uses) so that state dicts interchange with
        # torch.optim.Adam's: the reference saves and resumes optimizerG.pth / optimizerD.pth (train_gan.py:331-332,492-493)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self._tables = collections.OrderedDict()      # (param ptr, grad ptr)* -> _Table, least recently used first
        self._chunk = None
        self._arenas, self._arena_off = [], 0         # pinned staging; never freed (a captured graph re-reads its slices)
        self._free = {}                               # nbytes -> recycled pinned slices of evicted eager-mode tables

    MAX_TABLES = 8     # eager-mode bound: a new (param, grad) address pattern beyond this evicts the least recently used one

    def _pinned(self, nbytes):
        """slice of a pinned host arena.  Arenas are only ever added (a hipGraph that captured the upload of a table replays
        the copy FROM its pinned slice, so a slice that a capture has seen is never reused or freed), and only outside capture."""
        free = self._free.get(nbytes)
        if free:
            return free.pop()
        if not self._arenas or self._arena_off + nbytes > self._arenas[-1].numel():
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("HipAdam: pinned staging arena exhausted during graph capture; run a warm-up step first")
            total = sum(p.numel() for g in self.param_groups for p in g["params"])
            nparams = sum(len(g["params"]) for g in self.param_groups)
            per_table = 64 * nparams + 8 * (total // 4096 + nparams) + 256
            self._arenas.append(torch.empty(max(16 * per_table, nbytes), dtype=torch.uint8).pin_memory())
            self._arena_off = 0
        out = self._arenas[-1][self._arena_off: self._arena_off + nbytes]
        self._arena_off += nbytes
        return out

    def _state_for(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.zeros(1, dtype=torch.int32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
        return st

    def state_dict(self):
        """torch.optim.Adam's layout: per parameter ``step`` (f32 scalar on the host), ``exp_avg``, ``exp_avg_sq``."""
        sd = super().state_dict()
        sd["state"] = {k: {**st, "step": st["step"].detach().to("cpu", torch.float32).reshape(())} if "step" in st else st
                       for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        """Accepts a ``torch.optim.Adam`` state dict (a reference optimizerG.pth / optimizerD.pth) or our own."""
        for g in state_dict["param_groups"]:
            if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
                raise ValueError("HipAdam implements Adam without weight decay / amsgrad / maximize (train_gan.py:483-484)")
        super().load_state_dict(state_dict)
        for p, st in self.state.items():
            if "step" in st:
                st["step"] = torch.as_tensor(st["step"]).detach().reshape(1).round().to(device=p.device, dtype=torch.int32)
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st:
                    st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()
        self._tables.clear()          # the tables hold raw pointers into the replaced state tensors

    def _table(self, ps, device):
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
        hit = self._tables.get(key)
        if hit is not None:
            self._tables.move_to_end(key)
            return hit[:4]
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
        # pinned staging (allocated outside capture) + async copies: legal inside hipGraph capture
        raw = bytes(ents)
        chb = torch.tensor(chunks, dtype=torch.int32).numpy().tobytes()
        n0, n1 = (len(raw) + 63) // 64 * 64, (len(chb) + 63) // 64 * 64
        tab_h, ch_h = self._pinned(n0), self._pinned(n1)
        tab_h[: len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        ch_h[: len(chb)].copy_(torch.frombuffer(bytearray(chb), dtype=torch.uint8))
        tab = tab_h.to(device, non_blocking=True)
        ch = ch_h.to(device, non_blocking=True).view(torch.int32)
        capturing = torch.cuda.is_current_stream_capturing()
        self._tables[key] = (tab, ch, len(ps), len(chunks), tab_h, ch_h, capturing)
        # Gradients are re-allocated by every backward; when their addresses move (allocator churn at epoch boundaries,
        # evaluation, checkpointing) a new table is built.  Keep the working set bounded: evict the least recently used table
        # that no graph capture has seen and recycle its pinned slices.
        if len(self._tables) > self.MAX_TABLES and not capturing:
            torch.cuda.current_stream().synchronize()      # rare; the evicted table's upload must be done before its slice is reused
            for k in list(self._tables):
                if len(self._tables) <= self.MAX_TABLES:
                    break
                ent = self._tables[k]
                if k != key and not ent[6]:
                    del self._tables[k]
                    for h in (ent[4], ent[5]):
                        self._free.setdefault(h.numel(), []).append(h)
        return self._tables[key][:4]

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, scaler=None):
        """``grad_scale``: factor applied to every gradient element as the kernel reads it; 1.0 is torch.optim.Adam exactly.
        ``scaler`` (an `ops.LossScaler`, the IEEE-half mode): the gradients are those of scale x loss -- they are read times
        1 / scale; if ANY of them (over all parameter groups) is inf / NaN the step is skipped on the device (no parameter,
        moment or step counter changes) and the scale backs off; the scaler's counters say so afterwards."""
        assert closure is None
        groups = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            if not ps[0].is_cuda:
                raise RuntimeError("HipAdam runs on the GPU only (no CPU fallback)")
            groups.append((group, self._table(ps, ps[0].device)))
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream) if groups else None
        if scaler is None:
            for group, (tab, ch, nt, nc) in groups:
                b1, b2 = group["betas"]
                L.call("xmc_adam_step", C.c_void_p(tab.data_ptr()), nt, C.c_void_p(ch.data_ptr()), nc,
                       float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(grad_scale), st)
        else:
            assert grad_scale == 1.0
            CHECK, UPDATE, RESCALE = 1, 2, 4
            # every group is checked before any is updated; the last update also ends the scaler's step
            passes = [CHECK | UPDATE] if len(groups) == 1 else [CHECK, UPDATE]
            for mode in passes:
                for gi, (group, (tab, ch, nt, nc)) in enumerate(groups):
                    b1, b2 = group["betas"]
                    m = mode | (RESCALE if (mode & UPDATE) and gi == len(groups) - 1 else 0)
                    L.call("xmc_adam_step_scaled", C.c_void_p(tab.data_ptr()), nt, C.c_void_p(ch.data_ptr()), nc,
                           float(group["lr"]), float(b1), float(b2), float(group["eps"]), C.c_void_p(scaler.sf.data_ptr()),
                           C.c_void_p(scaler.si.data_ptr()), m, scaler.growth, scaler.backoff, scaler.interval, st)
        # the kernels wrote the parameters behind autograd's back: invalidate their packed copies and re-pack, in one launch,
        # the ones that exist (ops._PackEntry)
        changed = [p for group in self.param_groups for p in group["params"] if p.grad is not None]
        for p in changed:
            p._xmc_epoch = getattr(p, "_xmc_epoch", 0) + 1
        ops.repack_params(changed)
