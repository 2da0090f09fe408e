"""hipGraph replay of the whole G+D iteration.

The iteration enqueues ~1.5k short kernels; launched eagerly from Python the host becomes the bottleneck
(~55 ms per iteration regardless of batch).  Everything the step does is stream-ordered and free of host
synchronisation (kernels take device scalars for gamma / loss seeds, Adam keeps its step counters on the
device), so one iteration can be captured into a hipGraph once and replayed with new inputs copied into
static buffers.  Capture goes through ``torch.cuda.graph`` so PyTorch's caching allocator hands every
intermediate tensor a graph-private, replay-stable address.
"""
import torch

from . import ops


class GraphedIteration:
    """Callable with the signature of ``xmc_gan.train_gan.gan_iteration`` minus the modules.

    ``fn(imgs, sent_embs, words_embs, mask, noise) -> dict of 0-d loss tensors`` is captured after
    ``warmup`` eager iterations (which also populate the weight-pack / Adam-table caches).  One graph per
    N_CRITIC phase (whether the G step runs) is kept.
    """

    def __init__(self, step_fn, example_inputs, n_critic=1, warmup=3):
        self.step_fn = step_fn
        self.n_critic = max(1, int(n_critic))
        self.static_in = [t.clone() for t in example_inputs]
        self.graphs = {}
        self.static_out = {}
        self.it_state = {}
        self.warmup_left = warmup
        self.pool = None

    def _phase(self):
        return (self.it_state.get('i', 0) + 1) % self.n_critic == 0

    def __call__(self, *inputs):
        for s, t in zip(self.static_in, inputs):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        if self.warmup_left > 0:
            self.warmup_left -= 1
            return self.step_fn(*self.static_in, self.it_state)
        phase = self._phase()
        if phase not in self.graphs:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            state = dict(self.it_state)
            with torch.cuda.graph(g, pool=self.pool):
                out = self.step_fn(*self.static_in, state)
            self.pool = g.pool()
            self.graphs[phase] = g
            self.static_out[phase] = out
            ops.bump_weights_epoch()        # packed-weight cache entries now live in graph-private memory
        self.graphs[phase].replay()
        i = self.it_state.get('i', 0) + 1
        self.it_state['i'] = 0 if i % self.n_critic == 0 else i
        return self.static_out[phase]
