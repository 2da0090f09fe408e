"""Frozen text encoders with the reference's class names and call signature (model/encoder.py).

``RNN_ENCODER`` (the DAMSM caption encoder, encoder.py:73-153) holds an ``nn.Embedding`` and an ``nn.LSTM`` as parameter
containers -- ``state_dict()`` keys and shapes are upstream's, so ``text_encoder100.pth`` loads unchanged -- and runs
its forward on the MI355X kernels: embedding gather, one f32 MFMA GEMM for the input projections of every token and
both directions, and the per-sample LSTM recurrence kernel.  Forward only: the reference freezes the encoder and
puts it in eval mode (train_gan.py:464-468).

``SBERT_ENCODER`` (encoder.py:24-70) is a wrapper around the third-party ``sentence_transformers`` package and its
pretrained checkpoint; neither is part of this build, so constructing it fails loudly.
"""
import torch
import torch.nn as nn

from xmc_gan_amd import ops


class RNN_ENCODER(nn.Module):
    def __init__(self, cfg):
        super(RNN_ENCODER, self).__init__()
        self.n_steps = cfg.TEXT.MAX_LENGTH
        self.ntoken = cfg.TEXT.VOCA_SIZE
        self.ninput = 300
        self.drop_prob = 0.5
        self.nlayers = 1
        self.bidirectional = True
        self.rnn_type = cfg.TEXT.RNN_TYPE
        self.num_directions = 2
        self.nhidden = cfg.TEXT.EMBEDDING_DIM // self.num_directions
        if self.rnn_type not in ('LSTM', 'GRU'):
            raise NotImplementedError(f"TEXT.RNN_TYPE={self.rnn_type!r} (encoder.py:103)")
        self.ngates = 4 if self.rnn_type == 'LSTM' else 3
        self.encoder = nn.Embedding(self.ntoken, self.ninput)
        self.drop = nn.Dropout(self.drop_prob)
        # dropout= is a no-op for a single layer; left out to spare the construction-time warning (same parameters)
        rnn = nn.LSTM if self.rnn_type == 'LSTM' else nn.GRU               # encoder.py:95-102
        self.rnn = rnn(self.ninput, self.nhidden, self.nlayers, batch_first=True, bidirectional=True)
        self.encoder.weight.data.uniform_(-0.1, 0.1)                       # _init_weights (encoder.py:106-108)
        self.geom = ops.ConvGeom(self.ninput, 2 * self.ngates * self.nhidden, 1, 1, 0)
        self._packed = None

    def _weights(self):
        """[W_ih_fwd; W_ih_rev] (2*G*H, 300), input-side biases (2*G*H), [W_hh_fwd, W_hh_rev] (2, G*H, H) [, GRU: b_hn (2, H)];
        G = 4 gate rows (LSTM) or 3 (GRU); rebuilt when a parameter changes (load_state_dict, .to()).
        LSTM: the two bias vectors add up front.  GRU: b_hh of the r and z rows likewise, but the candidate gate is
        tanh(W_in x + b_in + r * (W_hn h + b_hn)), so its hidden bias stays with the recurrence."""
        r = self.rnn
        ps = (r.weight_ih_l0, r.weight_ih_l0_reverse, r.weight_hh_l0, r.weight_hh_l0_reverse,
              r.bias_ih_l0, r.bias_ih_l0_reverse, r.bias_hh_l0, r.bias_hh_l0_reverse)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is None or self._packed[0] != key:
            with torch.no_grad():
                w_ih = torch.cat((ps[0], ps[1]), 0).float().contiguous()
                w_hh = torch.stack((ps[2], ps[3]), 0).float().contiguous()
                if self.rnn_type == 'LSTM':
                    bias = torch.cat((ps[4] + ps[6], ps[5] + ps[7]), 0).float().contiguous()
                    extra = ()
                else:
                    H = self.nhidden
                    keep = torch.cat((torch.ones(2 * H), torch.zeros(H))).to(ps[6])            # b_hr, b_hz join; b_hn does not
                    bias = torch.cat((ps[4] + ps[6] * keep, ps[5] + ps[7] * keep), 0).float().contiguous()
                    extra = (torch.stack((ps[6][2 * H:], ps[7][2 * H:]), 0).float().contiguous(),)
            self._packed = (key, w_ih, bias, w_hh) + extra
        return self._packed[1:]

    def forward(self, caps, cap_lens, **kwargs):
        """caps int64 [B, n_steps] (0 = padding), cap_lens [B] -> words_embs [B, E, n_steps], sent_embs [B, E],
        mask [B, n_steps] (True at padding), all on the encoder's device."""
        if self.training:
            raise NotImplementedError("RNN_ENCODER runs frozen in eval mode (train_gan.py:466-468); the dropout of "
                                      "training mode (encoder.py:132) is not built")
        dev = self.encoder.weight.device
        caps, cap_lens = torch.as_tensor(caps), torch.as_tensor(cap_lens)
        if caps.dim() != 2 or caps.size(1) != self.n_steps:
            raise ValueError(f"caps must be [B, {self.n_steps}] token ids, got {tuple(caps.shape)}")
        if not caps.is_cuda:      # loader output: validate on the host for free (pack_padded_sequence raises upstream)
            if int(cap_lens.min()) < 1 or int(cap_lens.max()) > self.n_steps:
                raise ValueError("caption lengths must lie in [1, TEXT.MAX_LENGTH]")
            if int(caps.min()) < 0 or int(caps.max()) >= self.ntoken:
                raise IndexError("token id outside [0, TEXT.VOCA_SIZE)")
        caps = caps.to(dev, torch.int64, non_blocking=True)
        lens = cap_lens.to(dev, torch.int32, non_blocking=True).contiguous()
        w_ih, bias, w_hh, *extra = self._weights()
        B, T = caps.shape
        emb = ops.embedding(caps, self.encoder.weight)                                     # [B, T, 300]
        xproj = ops.linear(emb.view(B * T, self.ninput), w_ih, bias, self.geom, out_dtype=torch.float32)
        xproj = xproj.view(B, T, 2, self.ngates * self.nhidden)
        if self.rnn_type == 'LSTM':
            words_embs, sent_embs = ops.lstm_bidir(xproj, w_hh, lens, T)
        else:
            words_embs, sent_embs = ops.gru_bidir(xproj, w_hh, extra[0], lens, T)
        mask = (caps == 0)
        return words_embs, sent_embs, mask


class SBERT_ENCODER(nn.Module):
    def __init__(self, cfg):
        super(SBERT_ENCODER, self).__init__()
        raise ImportError(
            "SBERT_ENCODER wraps sentence_transformers.SentenceTransformer and its pretrained checkpoint "
            "(encoder.py:24-70); neither ships with this build.  Use a TEXT.ENCODER_NAME: 'RNN' preset or --synthetic.")
