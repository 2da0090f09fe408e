"""COCO caption/image datasets with the reference's class names, constructor arguments and item layout (dataset.py).

Host-side only (PIL + numpy; torchvision is not required): the loader hands CPU tensors to the training loop, which
moves them to the GPU (train_gan.py:183).  Item layout, as upstream: ``(img f32 [3,S,S] in [-1,1], [(caption, length)], key)``;
for ``TEXT.TYPE: 'WORD'`` the caption is an int64 vector of TEXT.MAX_LENGTH token ids padded with 0 (dataset.py:104-111), for
``'SENT'`` the raw sentence string and its word count (dataset.py:134-135).

Expected files under ``data/<DATASET_NAME>/`` (dataset.py:67-74,84-91,118-124): ``images/<key>.jpg``, ``<mode>/filenames.pickle``,
``captions.pickle`` = [train_caps, test_caps, i2w, w2i], ``bert_captions.pickle`` = [train_sents, test_sents].
"""
import os
import pickle

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


# ----------------------------------------------------------------------------- image transforms (PIL in, PIL out)
class Compose:
    def __init__(self, steps):
        self.steps = list(steps)

    def __call__(self, img):
        for f in self.steps:
            img = f(img)
        return img


class Resize:
    """int: shorter side -> size, aspect kept; (h, w): exact.  Bilinear, as torchvision.transforms.Resize on PIL images."""

    def __init__(self, size):
        self.size = size

    def __call__(self, img):
        w, h = img.size
        if isinstance(self.size, int):
            if (w <= h and w == self.size) or (h <= w and h == self.size):
                return img
            if w < h:
                nw, nh = self.size, int(self.size * h / w)
            else:
                nh, nw = self.size, int(self.size * w / h)
        else:
            nh, nw = self.size
        return img.resize((nw, nh), Image.BILINEAR)


class RandomCrop:
    """square crop at a uniformly random offset; draws (top, left) from torch's global RNG in torchvision's order."""

    def __init__(self, size):
        self.size = size

    def __call__(self, img):
        w, h = img.size
        s = self.size
        if h < s or w < s:
            raise ValueError(f"crop {s} larger than image {(h, w)}")
        top = int(torch.randint(0, h - s + 1, size=(1,)).item())
        left = int(torch.randint(0, w - s + 1, size=(1,)).item())
        return img.crop((left, top, left + s, top + s))


class RandomHorizontalFlip:
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img):
        if torch.rand(1) < self.p:
            return img.transpose(Image.FLIP_LEFT_RIGHT)
        return img


def to_normalized_tensor(img):
    """ToTensor + Normalize((0.5,)*3, (0.5,)*3) (dataset.py:34-37): uint8 HWC -> f32 CHW in [-1, 1]."""
    a = np.array(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(a.transpose(2, 0, 1).copy()).to(torch.float32).div_(255.0)
    return t.sub_(0.5).div_(0.5)


def train_transform(img_size):
    """train_gan.py:443-447"""
    return Compose([Resize(int(img_size * 76 / 64)), RandomCrop(img_size), RandomHorizontalFlip()])


def test_transform(img_size):
    """train_gan.py:454"""
    return Resize((img_size, img_size))


def get_img(img_path, normalize, transform=None):
    img = Image.open(img_path).convert('RGB')
    if transform is not None:
        img = transform(img)
    return normalize(img)


def index_to_sent(i2w_voca, caps):
    """token-id rows -> sentences, skipping the 0 padding (dataset.py:17-19)"""
    return [' '.join(i2w_voca[int(tok)] for tok in cap if int(tok) != 0) for cap in caps]


# ----------------------------------------------------------------------------- datasets
class TextDataset(Dataset):
    def __init__(self, data_dir, mode, transform, cfg):
        self.data_dir, self.mode, self.transform = data_dir, mode, transform
        self.img_size = cfg.IMG.SIZE
        self.b_local = False
        self.caps_per_image = cfg.TEXT.CAPTIONS_PER_IMAGE
        self.max_length = cfg.TEXT.MAX_LENGTH
        self.norm = to_normalized_tensor
        self.filenames = self._load_filenames(data_dir, mode)
        self._load_text_data(data_dir, mode)

    def __len__(self):
        return len(self.filenames)

    def __getitem__(self, idx):
        key = self.filenames[idx]
        img = get_img(f'{self.data_dir}/images/{key}.jpg', transform=self.transform, normalize=self.norm)
        sent_ix = 1                                           # upstream fixes the caption choice (dataset.py:49-50)
        texts = [self.get_caption(idx * self.caps_per_image + sent_ix)]
        if self.b_local:                                      # a second, different caption of the same image
            others = [k for k in range(self.caps_per_image) if k != sent_ix]
            texts.append(self.get_caption(idx * self.caps_per_image + int(np.random.choice(others))))
        return img, texts, key

    def _load_filenames(self, data_dir, mode):
        path = f'{data_dir}/{mode}/filenames.pickle'
        if not os.path.isfile(path):
            raise NotImplementedError(f'Download the meta data ({path} is missing)')
        with open(path, 'rb') as f:
            names = pickle.load(f)
        print(f'Load filenames from {path}, len : {len(names)}')
        return names

    def _load_text_data(self, data_dir, mode):
        raise NotImplementedError

    def get_caption(self, sent_ix):
        raise NotImplementedError


class WordTextDataset(TextDataset):
    """token-id captions for the RNN encoder"""

    def _load_text_data(self, data_dir, mode):
        path = os.path.join(data_dir, 'captions.pickle')
        if not os.path.isfile(path):
            raise NotImplementedError(f'{path} is missing')
        with open(path, 'rb') as f:
            train_caps, test_caps, i2w, w2i = pickle.load(f)[:4]
        print(f'Load from {path}, voca_size : {len(i2w)}')
        self.captions = train_caps if mode == 'train' else test_caps
        self.i2w, self.w2i, self.voca_size = i2w, w2i, len(i2w)

    def get_caption(self, sent_ix):
        toks = np.asarray(self.captions[sent_ix]).astype('int64')
        if (toks == 0).any():
            print('ERROR: do not need END (0) token', toks)
        n = min(len(toks), self.max_length)
        padded = np.zeros(self.max_length, dtype='int64')
        padded[:n] = toks[:n]
        return padded, n


class SentTextDataset(TextDataset):
    """raw sentences for the SBERT encoder"""

    def _load_text_data(self, data_dir, mode):
        path = os.path.join(data_dir, 'bert_captions.pickle')
        if not os.path.isfile(path):
            raise NotImplementedError(f'{path} is missing')
        with open(path, 'rb') as f:
            train_sents, test_sents = pickle.load(f)[:2]
        print(f'Load bert captions from {path}')
        self.captions = train_sents if mode == 'train' else test_sents

    def get_caption(self, sent_ix):
        s = self.captions[sent_ix]
        return s, len(s.split(' '))
