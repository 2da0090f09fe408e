"""Image dumps and scalar logging for the trainer's logging / evaluation tail (reference train_gan.py:150-160,297-326,338-395).

The reference leans on three packages that are optional here (none is on the step path): ``torchvision.utils.save_image`` for
the sample grids, ``tensorboard`` / ``wandb`` for scalars, ``pytorch_fid`` for the FID of the generated test set.  The grid
writer below restates what the reference's call ``save_image(x, path, normalize=True, scale_each=True)`` produces (eight images
per row, two pixels of padding, every image min-max scaled on its own) on PIL + numpy; the scalar log uses whichever backend is
importable and always keeps a JSON-lines copy next to it; FID is computed when ``pytorch_fid`` can be imported and reported as
unavailable otherwise."""
import json
import math
import os

import numpy as np


def make_grid(images, nrow=8, padding=2, normalize=True, scale_each=True, pad_value=0.0):
    """images: float array [B,C,H,W] (C = 1 or 3) -> float array [3, Hg, Wg] in [0,1] when normalised.
    Layout of torchvision.utils.make_grid: xmaps = min(nrow, B) images per row, `padding` pixels of `pad_value` around every
    image; `normalize` maps [min, max] to [0, 1] (per image with `scale_each`, else over the whole batch)."""
    x = np.asarray(images, dtype=np.float32)
    if x.ndim == 3:
        x = x[None]
    if x.shape[1] == 1:
        x = np.repeat(x, 3, axis=1)
    x = x.copy()
    if normalize:
        if scale_each:
            for i in range(x.shape[0]):
                lo, hi = float(x[i].min()), float(x[i].max())
                x[i] = np.clip((x[i] - lo) / max(hi - lo, 1e-5), 0.0, 1.0)
        else:
            lo, hi = float(x.min()), float(x.max())
            x = np.clip((x - lo) / max(hi - lo, 1e-5), 0.0, 1.0)
    B, C, H, W = x.shape
    if B == 1:
        return x[0]
    xmaps = min(nrow, B)
    ymaps = int(math.ceil(B / xmaps))
    h, w = H + padding, W + padding
    grid = np.full((C, h * ymaps + padding, w * xmaps + padding), pad_value, dtype=np.float32)
    k = 0
    for yy in range(ymaps):
        for xx in range(xmaps):
            if k >= B:
                break
            grid[:, yy * h + padding: yy * h + padding + H, xx * w + padding: xx * w + padding + W] = x[k]
            k += 1
    return grid


def save_image(images, path, nrow=8, padding=2, normalize=True, scale_each=True):
    """torchvision.utils.save_image(images, path, normalize=, scale_each=) (train_gan.py:160,299,326): grid -> 8-bit PNG."""
    from PIL import Image
    if hasattr(images, "detach"):
        images = images.detach().float().cpu().numpy()
    g = make_grid(images, nrow, padding, normalize, scale_each)
    arr = np.clip(g * 255.0 + 0.5, 0, 255).astype(np.uint8).transpose(1, 2, 0)
    Image.fromarray(arr).save(path)


class _Saver:
    """one background thread that turns queued sample batches into PNG grids (bounded queue: at most four batches wait on the host)"""

    def __init__(self):
        import queue
        import threading
        self.q, self.err = queue.Queue(maxsize=4), None
        self.t = threading.Thread(target=self._run, name="xmc-gan-image-writer", daemon=True)
        self.t.start()

    def _run(self):
        while True:
            args, kw = self.q.get()
            try:
                save_image(*args, **kw)
            except Exception as e:          # noqa: BLE001 -- reported by flush_saves() on the caller's thread
                self.err = e
            finally:
                self.q.task_done()


_saver = None


def save_image_async(images, path, **kw):
    """`save_image` off the training loop's thread.  Grid + PNG encoding of 256 samples at 256 px is ~1.9 s of host time (an 8 258 x 2 066
    image through zlib), every LOG_INTERVAL steps: written inline it cost a 1 000-iteration run 15 % of its throughput (50.9 against 43 ms
    per iteration).  The device-to-host copy is made HERE, before returning (under graph replay the tensor is a static output the next
    iteration overwrites); normalisation, grid and encoding run on a writer thread (numpy and zlib release the interpreter lock).
    `flush_saves()` waits for the files."""
    global _saver
    if hasattr(images, "detach"):
        host = images.detach().float().cpu()
        if host.data_ptr() == images.data_ptr():        # already an f32 host tensor: .cpu() handed back the caller's memory
            host = host.clone()
        images = host.numpy()
    else:
        images = np.array(images, dtype=np.float32, copy=True)
    if _saver is None:
        _saver = _Saver()
    _saver.q.put(((images, path), kw))


def flush_saves():
    """block until every queued image file is on disk; re-raises what the writer thread caught"""
    if _saver is None:
        return
    _saver.q.join()
    if _saver.err is not None:
        e, _saver.err = _saver.err, None
        raise e


def to_uint8_hwc(img):
    """one generated / real image in [-1, 1], [3,H,W] -> uint8 [H,W,3] the way eval() writes them ((x + 1) * 127.5, truncated;
    train_gan.py:366-379)"""
    if hasattr(img, "detach"):
        img = img.detach().float().cpu().numpy()
    return np.transpose(((np.asarray(img, dtype=np.float32) + 1.0) * 127.5).astype(np.uint8), (1, 2, 0))


class ScalarLog:
    """add_scalar(tag, value, step) onto TensorBoard (``--log_type tb``, train_gan.py:312-320) or wandb (``--log_type wdb``,
    300-311) when the package imports, and always onto ``<log_dir>/scalars.jsonl``."""

    def __init__(self, log_dir, log_type="tb", run_name=None):
        self.backend, self._tb, self._wb, self._pending, self._step = "jsonl", None, None, {}, None
        self._f = open(os.path.join(log_dir, "scalars.jsonl"), "a") if log_dir else None
        if log_type == "wdb":
            try:
                import wandb
                wandb.init(project="xmc_gan", name=run_name)
                self._wb, self.backend = wandb, "wandb"
            except Exception:       # noqa: BLE001 -- not installed / no network: the JSON-lines copy remains
                pass
        elif log_dir:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self._tb, self.backend = SummaryWriter(log_dir), "tensorboard"
            except Exception:       # noqa: BLE001
                pass

    def add_scalar(self, tag, value, step):
        value = float(value)
        if self._f:
            self._f.write(json.dumps({"tag": tag, "value": value, "step": int(step)}) + "\n")
            self._f.flush()
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)
        if self._wb is not None:       # one wandb.log per step, like the reference's log_dict
            if self._step is not None and step != self._step:
                self.flush()
            self._step = step
            self._pending[tag] = value

    def flush(self):
        if self._wb is not None and self._pending:
            self._wb.log(dict(self._pending))
            self._pending.clear()
        if self._tb is not None:
            self._tb.flush()

    def close(self):
        self.flush()
        if self._tb is not None:
            self._tb.close()
        if self._f:
            self._f.close()


def fid_between(dir_a, dir_b, device, batch_size=100, dims=2048):
    """calculate_fid_given_paths([org_dir, save_dir], ...) (train_gan.py:389) when pytorch_fid imports; None otherwise."""
    try:
        from pytorch_fid.fid_score import calculate_fid_given_paths
    except Exception:       # noqa: BLE001 -- optional third-party package (and its pretrained Inception weights)
        return None
    return float(calculate_fid_given_paths([dir_a, dir_b], batch_size=batch_size, device=device, dims=dims))
