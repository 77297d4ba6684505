"""The Gaussian toy game agents (BASELINE config 1) on the engine: host-side mirror of the reference's
``models/gim_gaussian_models.py`` (GIMGaussianDis :17, GIMGaussianAuthenticator :47, GIMGaussianImpersonator :66,
get_im :94, get_au :101) with the same state-dict keys.  Pure MLP plumbing: linears, set statistics, noise combine -
the same kernels as the image path's head.  ``forward`` of the impersonator takes an optional ``z`` (drawn at :81)."""
import torch
import torch.nn as nn

from . import model_blocks as mb
from . import ops


class GIMMeanStdStat(nn.Module):
    """models/gim_basic_models.py:71-89 (fused with the concat in GIMGaussianDis.forward)."""

    def __init__(self):
        super().__init__()
        self.n_stats = 2


class GIMGaussianDis(nn.Module):
    def __init__(self, src_dim, stat):
        super().__init__()
        self.src_dim = src_dim
        self.stat = stat
        self.n_stats = stat.n_stats
        self.mlp = mb.MLP((self.n_stats * src_dim * 2, src_dim, 2 * src_dim, 1))
        for m in self.mlp.model:  # weights_init('kaiming'), models/model_blocks.py:18-38
            if isinstance(m, mb.GimLinear):
                nn.init.kaiming_normal_(m.weight.data, a=0.2)
                nn.init.constant_(m.bias.data, 0.0)

    def forward(self, test_sample, si_sample):
        return self.mlp(ops.mean_std_cat(test_sample, si_sample))


class GIMGaussianAuthenticator(nn.Module):
    def __init__(self, dis):
        super().__init__()
        self.dis = dis

    def forward(self, test_sample, si_sample):
        return self.dis(test_sample=test_sample, si_sample=si_sample)


class GIMGaussianImpersonator(nn.Module):
    def __init__(self, src_dim, env_noise_mapper):
        super().__init__()
        self.src_dim = src_dim
        self.env_noise_mapper = env_noise_mapper
        self.out_mlp = mb.MLP((2 * src_dim, 2 * src_dim, src_dim))  # constructed, never used (as in the reference)

    def forward(self, leaked_sample, n, remove_noise_mean=True, z=None):
        batch_size = leaked_sample.size(0)
        src = ops.mean_dim1(leaked_sample)
        if z is None:
            z = torch.randn((batch_size, n, self.src_dim), device=leaked_sample.device)
        w = self.env_noise_mapper(z)
        return ops.noise_combine(src, w, remove_noise_mean)


def get_im(src_dim):
    return GIMGaussianImpersonator(src_dim=src_dim, env_noise_mapper=mb.MLP([src_dim, src_dim]))


def get_au(src_dim):
    return GIMGaussianAuthenticator(dis=GIMGaussianDis(src_dim=src_dim, stat=GIMMeanStdStat()))
