"""The Gaussian toy game's caller loop (BASELINE config 1) on the engine: host-side mirror of the reference's
``training/gim_gaussian_training.py`` - ``im_train_step`` (:20-30), ``au_train_step`` (:33-47), ``train`` (:50-151),
``train_gim_gaussian`` (:154-235) - with the same signatures, logger keys and cadences.

Sampling: per iteration the source means ``mu ~ N(0, prior_sigma)`` and then the real, leaked and registration sets
``~ N(mu, src_sigma)``, in that order, all from torch's DEFAULT HOST generator, then moved to the device.  That is the
reference's stream when it runs on a CPU (:72-86) - the case BASELINE config 1 names and the parity fixture pins.  (On
``cuda`` the reference moves ``mu`` to the device first, so there its three sample sets come from the DEVICE generator; the
engine keeps the host draws on the GPU too: one documented stream instead of two.)  ``host_noise=True`` also draws the
impersonator's latent ``z`` (models/gim_gaussian_models.py:81) from the host generator right behind them - what a CPU run of
the reference consumes; by default ``z`` is drawn on the device.

Data parallel (``EpisodeParallel``, torch.distributed initialised): every rank draws the SAME global batch of ``batch_size``
episodes - the ranks must seed the host generator identically, as ``train_gim_on_gaussians.py:6`` does with
``torch.manual_seed`` - and keeps its own slice (``trainer.shard``): ONE batch of ``batch_size`` scattered over the GPUs, what
``nn.DataParallel`` does in the reference (:198-201), not ``world * batch_size``.
"""
import os

import torch

from . import model_blocks as mb
from .gim_gaussian_trainer import GIMGaussianTrainer
from .gim_img_training import au_train_step, gim_step, im_train_step  # noqa: F401  (same step protocol as the image game)
from .training_logger import Logger
from .training_utils import DataParallelMock, EpisodeParallel

_TRAIN_KEYS = (('train losses', 'au loss'), ('train losses', 'au loss on real'), ('train losses', 'au loss on fake'),
               ('train losses', 'au reg'), ('train au out', 'au out on real'), ('train au out', 'au out on fake'))


def _draw_sets(batch_size, src_dim, src_sigma, prior_sigma, sizes, device):
    """mu [B, d] and one sample set [B, t, d] per entry of `sizes`, drawn on the host in the reference's order."""
    f32 = torch.float32    # the engine's dtype: the draws are float32 draws whatever the process-wide default dtype is
    mu = torch.normal(mean=torch.zeros((batch_size, src_dim), dtype=f32), std=torch.full((batch_size, src_dim), float(prior_sigma), dtype=f32))
    sigma = torch.full((batch_size, src_dim), float(src_sigma), dtype=f32)
    sets = [torch.normal(mean=mu.unsqueeze(1).repeat(1, t, 1), std=sigma.unsqueeze(1).repeat(1, t, 1)) for t in sizes]
    return mu.to(device), sigma.to(device), [s.to(device) for s in sets]


def _l1(a, b):
    return float((a - b).abs().mean())


def train(device, trainer, logger, n_iters, batch_size, src_dim, src_sigma, prior_sigma, save_stats_every, save_every,
          host_noise=False):
    mod = trainer.module
    for _ in range(n_iters):
        mod.do_global_step()
        m, n, k = mod.m, mod.n, mod.k
        mu, sigma, (real_sample, leaked_sample, si_sample) = _draw_sets(batch_size, src_dim, src_sigma, prior_sigma, (n, m, k), device)
        z = torch.randn((batch_size, n, src_dim), dtype=torch.float32).to(device) if host_noise else None
        if getattr(trainer, "world_size", 1) > 1:   # one global batch, scattered over the ranks (identical host seeds: see above)
            mu, sigma, real_sample, leaked_sample, si_sample = trainer.shard(mu, sigma, real_sample, leaked_sample, si_sample)
            z = trainer.shard(z) if z is not None else None
        global_step = mod.get_global_step()

        im_res, au_res = gim_step(trainer, leaked_sample, real_sample, si_sample, z=z)
        fake_sample = au_res[8]
        # one host round trip for the iteration's statistics
        acc_real = au_res[6].to(torch.float).mean()
        acc_fake = torch.eq(au_res[7], 0).to(torch.float).mean()
        vals = torch.stack([im_res[0].reshape(())] + [t.reshape(()) for t in au_res[:6]] + [0.5 * (acc_real + acc_fake), acc_real, acc_fake]).tolist()
        logger.add_scalar(category='train losses', k='im loss', v=vals[0], global_step=global_step)
        for (cat, key), v in zip(_TRAIN_KEYS, vals[1:7]):
            logger.add_scalar(category=cat, k=key, v=v, global_step=global_step)
        for key, v in zip(('au acc', 'au acc on real', 'au acc on fake'), vals[7:]):
            logger.add_scalar(category='train accuracy', k=key, v=v, global_step=global_step)

        if global_step % save_stats_every == 0:
            with torch.no_grad():
                for cat, sample, with_leak in (('im distances', fake_sample, True), ('real distances', real_sample, False)):
                    mean = sample.mean(dim=1)
                    if with_leak:
                        logger.add_scalar(category=cat, k='l1_dist_from_leaked_sample_mean', v=_l1(mean, leaked_sample.mean(dim=1)),
                                          global_step=global_step)
                    logger.add_scalar(category=cat, k='l1_dist_from_gt_sample_mean', v=_l1(mean, mu), global_step=global_step)
                    logger.add_scalar(category=cat, k='l1_dist_from_gt_std', v=_l1(mb.custom_std(sample), sigma), global_step=global_step)
        if global_step % save_every == 0:
            mod.save()


def train_gim_gaussian(device_name, device_ids, outdir, authenticator, impersonator, m, n, k, src_dim, src_sigma, prior_sigma,
                       reg_param, remove_noise_mean, au_lr, im_lr, resume_from_ckpt, n_iters, batch_size, save_every, save_stats_every):
    if device_name != 'cuda':
        raise RuntimeError("the GIM engine has no CPU compute path: device_name must be 'cuda' (an MI355X)")
    assert batch_size % len(device_ids) == 0
    device = torch.device('cuda', torch.cuda.current_device())
    logger = Logger(log_dir=os.path.join(outdir, 'logs'), img_dir=os.path.join(outdir, 'imgs'), tensorboard_dir=os.path.join(outdir, 'tb'))
    trainer = GIMGaussianTrainer(outdir=outdir, m=m, n=n, k=k, authenticator=authenticator.to(device), impersonator=impersonator.to(device),
                                 au_lr=au_lr, im_lr=im_lr, reg_param=reg_param, remove_noise_mean=remove_noise_mean).to(device)
    if resume_from_ckpt:
        trainer.resume_from_ckpt(ckpt_path=resume_from_ckpt)
    # one process per GPU (EpisodeParallel) replaces nn.DataParallel (:198-201); a single process is the mock wrapper
    trainer = EpisodeParallel(trainer) if torch.distributed.is_available() and torch.distributed.is_initialized() else DataParallelMock(trainer)
    try:
        train(device=device, trainer=trainer, logger=logger, n_iters=n_iters, batch_size=batch_size, src_dim=src_dim, src_sigma=src_sigma,
              prior_sigma=prior_sigma, save_stats_every=save_stats_every, save_every=save_every)
    except (KeyboardInterrupt, PermissionError) as e:
        print("\n%s\n%s\nSaving checkpoint...\n" % (type(e).__name__, e))
        trainer.module.save()
    logger.save_stats('stats.p')
