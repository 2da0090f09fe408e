"""hipGraph replay of the whole G+D iteration, with eager seams for the collectives.

The iteration enqueues ~1 k short kernels; launched eagerly from Python the host becomes the bottleneck (~55 ms per iteration
regardless of batch).  Everything the step does is stream-ordered and free of host synchronisation (kernels take device scalars
for gamma / loss seeds, Adam keeps its step counters on the device), so it can be captured once and replayed with new inputs
copied into static buffers.  PyTorch's caching allocator hands every intermediate tensor a graph-private, replay-stable address.

Data parallel: a collective cannot live inside the capture (gloo has no device-side form, and an RCCL launch inside a graph
ties the replay to one communicator state), so the iteration is captured as a SEQUENCE of graphs that share one memory pool,
cut at every collective: `seam(fn)` ends the running capture, runs `fn` (the all-reduce / all-gather on tensors that live in
the pool, hence at fixed addresses) eagerly on the same stream, and opens the next capture.  A replay launches the graphs and
calls the recorded collectives in the recorded order: 3-4 graph launches and 2-3 collective calls of host work per iteration
instead of ~1 k kernel launches, for any number of ranks.
"""
import time

import torch

from . import ops

_active = None          # the GraphedIteration whose capture is running (collectives consult it through seam())


def _quiesce():
    """Called outside capture, before a capture is opened or re-opened: nothing of a collective may be pending anywhere.  The `nccl` process
    group's watchdog thread polls the end event of every collective it has not yet seen complete (every 100 ms); on this ROCm such a poll
    WHILE this thread captures fails now and then with "operation not permitted on an event last recorded in a capturing stream" -- the
    watchdog throws, the process aborts, or the capture is invalidated (RCCL at world size 1, 64 px, batch 32: about 1 run in 10, with the
    collectives issued synchronously or not; `capture_error_mode="thread_local"` does not help, the failing call is the other thread's).  So
    the device is drained and the watchdog is given two polling periods to retire what it holds: it has nothing to poll while the capture
    runs.  Once per seam of a capture, never at replay."""
    if not torch.cuda.is_available():
        return
    torch.cuda.synchronize()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        time.sleep(0.25)


def seam(fn):
    """Run the collective ``fn()`` now; under capture: between two graphs, and again at every replay."""
    if _active is None:
        return fn()
    return _active._seam(fn)


class GraphedIteration:
    """Callable with the signature of ``xmc_gan.train_gan.gan_iteration`` minus the modules.

    ``fn(imgs, sent_embs, words_embs, mask, noise) -> dict of 0-d loss tensors`` is captured after
    ``warmup`` eager iterations (which also populate the weight-pack / Adam-table caches).  One sequence per
    N_CRITIC phase (whether the G step runs) is kept.
    """

    def __init__(self, step_fn, example_inputs, n_critic=1, warmup=3):
        self.step_fn = step_fn
        self.n_critic = max(1, int(n_critic))
        self.static_in = [t.clone() for t in example_inputs]
        self.seqs = {}          # phase -> [callable, ...]  (graph.replay and collectives, in order)
        self.static_out = {}
        self.it_state = {}
        self.warmup_left = warmup
        self.pool = None
        self.stream = None
        self._graphs = []       # keep the CUDAGraph objects alive
        self._cur = None
        self._seq = None
        self._pack_serial = None    # ops' pack-entry counter when the last capture ended
        self._counted = 0           # captures ops counts as alive on our behalf
        self.seam_host_s = [0.0, 0]  # host time spent inside the collectives' calls at replay [seconds, calls] (bench.py `dist`)

    def close(self):
        """drop the graphs; lets ops release the packed-weight buffers it kept alive for their replays"""
        self.seqs, self.static_out, self._graphs = {}, {}, []
        while self._counted > 0:
            self._counted -= 1
            ops.graph_released()

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass

    def seams_per_iteration(self):
        """collective calls a replay makes, per captured phase (0 without data parallelism)"""
        return {("g_step" if ph else "d_only"): sum(1 for it in seq if getattr(it, "__name__", "") == "eager")
                for ph, seq in self.seqs.items()}

    def _phase(self):
        return (self.it_state.get('i', 0) + 1) % self.n_critic == 0

    # ---- capture plumbing (what torch.cuda.graph does, split so that a capture can be ended and re-opened)
    def _begin(self):
        g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        # thread_local: calls made by OTHER threads while this one captures (the process group's watchdog polling its events)
        # must not invalidate the capture
        g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
        self._cur = g

    def _end(self):
        self._cur.capture_end()
        self._graphs.append(self._cur)
        self._seq.append(self._cur.replay)
        self._cur = None

    def _seam(self, fn):
        acc = self.seam_host_s

        def eager():                 # replays run outside the autograd context the collective was first issued in
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("a collective was issued while its stream is capturing")
            t0 = time.perf_counter()
            with torch.no_grad():
                out = fn()
            acc[0] += time.perf_counter() - t0
            acc[1] += 1
            return out
        self._end()
        out = eager()
        _quiesce()
        self._seq.append(eager)
        self._begin()
        return out

    def _capture(self, phase):
        global _active
        import gc
        torch.cuda.synchronize()
        gc.collect()
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        state = dict(self.it_state)
        self._seq = []
        ok = False
        with torch.cuda.stream(self.stream):
            _active = self
            try:
                _quiesce()          # (the warm-up iterations' collectives)
                self._begin()
                out = self.step_fn(*self.static_in, state)
                self._end()
                ok = True
            finally:
                _active = None
                if self._cur is not None:           # an exception inside the step: leave capture mode before it propagates
                    try:
                        self._cur.capture_end()
                    except Exception:
                        pass
                    self._cur = None
                if not ok:
                    # nothing recorded by the aborted capture ever ran: packed weights it created are uninitialised memory and
                    # its re-pack marked others valid without re-packing them.  Invalidate all of it before anyone runs eagerly.
                    self._seq = None
                    ops.end_of_capture(failed=True)
        torch.cuda.current_stream().wait_stream(self.stream)
        self.seqs[phase] = self._seq
        self.static_out[phase] = out
        self._seq = None
        # packed-weight entries created inside the capture live in graph-private memory
        self._pack_serial = ops.end_of_capture()
        self._counted += 1

    def __call__(self, *inputs):
        for s, t in zip(self.static_in, inputs):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        if self.warmup_left > 0:
            self.warmup_left -= 1
            return self.step_fn(*self.static_in, self.it_state)
        phase = self._phase()
        if phase not in self.seqs:
            self._capture(phase)
        for item in self.seqs[phase]:
            item()
        # the replay changed the weights without the host-side bookkeeping of an eager optimizer step: packed copies made since
        # the capture were not refreshed by it
        self._pack_serial = ops.drop_packs_newer_than(self._pack_serial)
        i = self.it_state.get('i', 0) + 1
        self.it_state['i'] = 0 if i % self.n_critic == 0 else i
        return self.static_out[phase]
