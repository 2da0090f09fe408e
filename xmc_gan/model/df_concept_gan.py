"""Attention-modulation generators (reference model/df_concept_gan.py) -- placeholder until the grouped-conv /
region-attention kernels land; class names and the NetD contract are already in place."""
import torch.nn as nn

from .df_gan import gen_arch, disc_arch  # noqa: F401  (same tables upstream: df_concept_gan.py:10-62)


class InNetG(nn.Module):
    def __init__(self, cfg, **kwargs):
        super().__init__()
        raise NotImplementedError("CONCEPT_IN_DF_GEN: HIP attention-modulation blocks not built yet")


class OutNetG(nn.Module):
    def __init__(self, cfg, **kwargs):
        super().__init__()
        raise NotImplementedError("CONCEPT_OUT_DF_GEN: HIP attention-modulation blocks not built yet")


class NetD(nn.Module):
    def __init__(self, cfg, **kwargs):
        super(NetD, self).__init__()
        raise NotImplementedError      # as upstream (df_concept_gan.py:587)
