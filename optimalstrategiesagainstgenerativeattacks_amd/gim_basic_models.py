"""Set-pooling statistics of the authenticator head (mirror of the used subset of the reference's
``models/gim_basic_models.py``: GIMMeanStat :20, GIMStdStat :37, GIMFCStat :113, GIMMeanStdFcStat :152).

On the hot path the three statistics of both sample sets and the concat to the [B, 5120] head input are
ONE fused operator (``ops.head_cat``); the classes below keep the reference's module tree (state-dict key
``fc.stat.model.*``), expose the per-sample FC features that operator needs, and implement every ``forward`` of
the reference on the same kernels (``gim_set_stats_fwd / _bwd``).
"""
import torch.nn as nn

from . import model_blocks as mb
from . import ops


class GIMMeanStat(nn.Module):
    def __init__(self):
        super().__init__()
        self.n_stats = 1

    def forward(self, x):
        return ops.mean_dim1(x)


class GIMStdStat(nn.Module):
    def __init__(self):
        super().__init__()
        self.n_stats = 1

    def forward(self, x):
        """models/gim_basic_models.py:37-51: sqrt(unbiased variance over the set + 1e-8), zeros for a one-element set."""
        return mb.custom_std(x)


class GIMFCStat(nn.Module):
    def __init__(self, style_dim, n_stats=1, hidden_layers=()):
        super().__init__()
        self.style_dim = style_dim
        self.n_stats = n_stats
        self.fc_layer_dims = [style_dim] + [*hidden_layers] + [n_stats * style_dim]
        self.stat = mb.MLP(self.fc_layer_dims)
        self.sample_mean = GIMMeanStat()

    def per_sample(self, x):
        """MLP(x) for every element of the set: [B, t, D] -> [B, t, n_stats*D]."""
        return self.stat(x)

    def forward(self, x):
        return self.sample_mean(self.stat(x))


class GIMMeanStdFcStat(nn.Module):
    def __init__(self, style_dim, fc_n_stats, fc_hidden_layers):
        super().__init__()
        self.n_stats = 2 + fc_n_stats
        self.sample_mean = GIMMeanStat()
        self.sample_std = GIMStdStat()
        self.fc = GIMFCStat(style_dim=style_dim, n_stats=fc_n_stats, hidden_layers=fc_hidden_layers)

    def forward(self, x):
        """models/gim_basic_models.py:169-172: cat(mean, std, mean of the FC features) of ONE sample set [B, t, D] -> [B, n_stats*D]
        (one fused operator; GIMFaceDis.forward uses the two-set form ops.head_cat, which also writes the source means)."""
        return ops.stat_cat(x, self.fc.per_sample(x))
