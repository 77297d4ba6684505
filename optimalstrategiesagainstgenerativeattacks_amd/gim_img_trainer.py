"""GIMImgTrainer on the MI355X engine: host-side mirror of the reference's ``training/gim_img_trainer.py``
(constructor, the three ``forward`` modes and their return tuples, optimizer / scheduler attributes,
checkpoint hooks, lr / global-step accessors).

Differences that do not change any result of the training loop:
  * both optimizers are ``FusedAdam`` (one flat-buffer kernel per step, RCCL all-reduce built in);
  * in ``impersonator_forward`` the authenticator's parameters are frozen for the duration of the forward, so
    the generator step does not compute the discriminator weight gradients that the reference computes and
    then discards (``authenticator_opt.zero_grad()`` at training/gim_img_training.py:172) - SURVEY.md 8(a);
  * the R1 term (``reg_param > 0``) differentiates the authenticator's input gradient through the same HIP kernels
    (ops.ConvDgradFn and friends); no torch operator is on that path either.
"""
import os
from contextlib import contextmanager

import torch
import torch.nn as nn
import torch.optim as optim

from . import ops
from .optim import FusedAdam
from .training_utils import CheckpointIO, GlobalStep, compute_grad2, num_parameters


@contextmanager
def _frozen(module):
    ps = [p for p in module.parameters() if p.requires_grad]
    for p in ps:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p in ps:
            p.requires_grad_(True)


class GimTrainerBase(nn.Module):
    """What the image and the Gaussian trainer share: the mode-string forward, the BCE game losses, the assembly of the
    discriminator step's 9-tuple, checkpoint wiring and the global-step counter (training/gim_img_trainer.py:79-94,158-172,
    training/gim_gaussian_trainer.py - the reference writes these twice)."""
    CHECKPOINT_DIR = "ckpts"
    MODES = ("authenticator_forward", "impersonator_forward", "impersonator_sample")
    CHECKPOINT_KEYS = ("authenticator", "impersonator", "authenticator_opt", "impersonator_opt")

    def _init_agents(self, m, n, k, authenticator, impersonator, reg_param, remove_noise_mean):
        self.m, self.n, self.k = m, n, k
        self.authenticator, self.impersonator = authenticator, impersonator
        self.reg_param, self.remove_noise_mean = reg_param, remove_noise_mean
        self._global_step = GlobalStep()

    def _init_checkpoints(self, outdir):
        for label, net in (("Authenticator", self.authenticator), ("impersonator", self.impersonator)):
            print("{} has {} parameters".format(label, num_parameters(net.parameters())))
        self.checkpoint_dir = os.path.join(outdir, self.CHECKPOINT_DIR)
        self.checkpoint_io = CheckpointIO(checkpoint_dir=self.checkpoint_dir)
        self.checkpoint_io.register_modules(**{key: getattr(self, key) for key in self.CHECKPOINT_KEYS},   # the dict keys of the .pt file
                                            global_step=self._global_step)

    def forward(self, mode, **kwargs):
        """One entry point with a mode string, so that a data-parallel wrapper sees a plain nn.Module.forward."""
        if mode not in self.MODES:
            raise ValueError("unsupported mode")
        return getattr(self, mode)(**kwargs)

    def gan_loss(self, dis_out, target, reduce=False):
        """BCE-with-logits against a constant target, per episode (training/gim_img_trainer.py:90-94)."""
        loss = ops.bce_logits(dis_out, float(target))
        return (loss.mean() if reduce else loss).squeeze()

    def _discriminator_terms(self, out_on_real, out_on_fake_fn, real_sample, si_sample, fake_sample, grad):
        """The 9-tuple of authenticator_forward from the logits on the real sample and a callable for the logits on the fake one
        (called AFTER the R1 term, as the reference orders its two discriminator calls: training/gim_img_trainer.py:102-123)."""
        loss_on_real = self.gan_loss(dis_out=out_on_real, target=1.)
        if grad and self.reg_param > 0:
            reg = self.reg_param * compute_grad2(out_on_real, (real_sample, si_sample))
        else:
            reg = torch.zeros_like(loss_on_real)
        out_on_fake = out_on_fake_fn()
        loss_on_fake = self.gan_loss(dis_out=out_on_fake, target=0.)
        with torch.no_grad():
            pred_on_real, pred_on_fake = out_on_real.detach() >= 0, out_on_fake.detach() >= 0
        return (loss_on_real + loss_on_fake + reg, loss_on_real.detach(), loss_on_fake.detach(), reg, out_on_real.detach(),
                out_on_fake.detach(), pred_on_real, pred_on_fake, fake_sample.detach())

    def _wants_input_grad(self, real_sample, si_sample):
        if self.reg_param > 0:   # R1 differentiates the logits w.r.t. the discriminator's inputs (training/gim_img_trainer.py:98-100)
            real_sample.requires_grad_()
            si_sample.requires_grad_()

    def impersonator_sample(self, leaked_sample, z=None):
        with torch.no_grad():
            return self.impersonator(leaked_sample=leaked_sample, n=self.n, remove_noise_mean=self.remove_noise_mean, z=z)

    def resume_from_ckpt(self, ckpt_path):
        self.checkpoint_io.load(ckpt_path)
        print("Resuming training from iteration {}".format(self.global_step))

    def get_global_step(self):
        return self._global_step.get()

    def do_global_step(self):
        return self._global_step.step()

    @property
    def global_step(self):
        return self.get_global_step()


class GIMImgTrainer(GimTrainerBase):
    IM_GROUPS = ("src_encoder", "env_encoder", "env_decoder", "img2img", "img_att", "env_noise_mapper")

    def __init__(self, outdir, m, n, k, authenticator, impersonator, au_lr, im_lr, env_noise_mapping_lr,
                 beta1=0., beta2=0.99, lr_milestones=(), lr_gamma=0.3, reg_param=10., remove_noise_mean=True):
        super().__init__()
        self._init_agents(m, n, k, authenticator, impersonator, reg_param, remove_noise_mean)
        betas = (beta1, beta2)
        self.authenticator_opt = FusedAdam(self.authenticator.parameters(), lr=au_lr, betas=betas)
        # six parameter groups in the reference's order (their indices are part of the checkpoint format,
        # training/gim_img_trainer.py:52-58); only the noise mapper has a learning rate of its own
        groups = [(sub, env_noise_mapping_lr if sub == "env_noise_mapper" else im_lr) for sub in self.IM_GROUPS]
        self.impersonator_opt = FusedAdam([{"params": getattr(self.impersonator, sub).parameters(), "lr": lr} for sub, lr in groups],
                                          lr=im_lr, betas=betas)
        self.au_scheduler, self.im_scheduler = (self.get_lr_scheduler(optimizer=o, milestones=lr_milestones, gamma=lr_gamma)
                                                for o in (self.authenticator_opt, self.impersonator_opt))
        self._init_checkpoints(outdir)

    def authenticator_forward(self, fake_sample, real_sample, si_sample, grad=True):
        ops.join_lanes()
        self._wants_input_grad(real_sample, si_sample)
        au = self.authenticator
        au.prefetch_spectral(3)  # si, real, fake: three calls of each encoder
        # same per-encoder call order as the reference (si, real, fake); the two encoders run on two streams
        (si_src, real_src, fake_src), (si_env, real_env, fake_env) = au.encode_samples([si_sample, real_sample, fake_sample])
        return self._discriminator_terms(
            au.dis(test_src=real_src, test_env=real_env, si_src=si_src, si_env=si_env),
            lambda: au.dis(test_src=fake_src, test_env=fake_env, si_src=si_src, si_env=si_env),
            real_sample, si_sample, fake_sample, grad)

    def impersonator_forward(self, leaked_sample, si_sample, z=None):
        fake_sample = self.impersonator(leaked_sample=leaked_sample, n=self.n, remove_noise_mean=self.remove_noise_mean, z=z)
        ops.join_lanes()   # a discriminator step still running on lane 1 (gim_step(defer_join=True)) owns the weights read next
        with _frozen(self.authenticator):
            auth_out = self.authenticator(test_sample=fake_sample, si_sample=si_sample)
        return self.gan_loss(dis_out=auth_out, target=1.), fake_sample, auth_out

    # ---- checkpoints (format: training_utils.CheckpointIO; file name as training/gim_img_trainer.py:158-172) ----
    def save(self, epoch):
        ops.join_lanes()   # the discriminator step of the last iteration may still be running on lane 1
        step = self.global_step
        print("\nSaving checkpoint...\n")
        self.checkpoint_io.save(global_step=step, last_epoch=epoch, filename="model_{:08}.pt".format(step))

    # ---- learning rates: MultiStepLR over both optimizers, stepped once per iteration by the caller loop ----
    def get_lr_scheduler(self, optimizer, milestones, gamma):
        return optim.lr_scheduler.MultiStepLR(optimizer=optimizer, milestones=list(milestones), gamma=gamma,
                                              last_epoch=self.global_step)

    def update_learning_rate(self):
        for sched in (self.au_scheduler, self.im_scheduler):
            if sched is not None:
                sched.step()

    @property
    def au_lr(self):
        return self.au_scheduler.get_last_lr()[0]

    @property
    def im_lr(self):
        return self.im_scheduler.get_last_lr()[0]

    @property
    def im_noise_mapping_lr(self):
        return self.im_scheduler.get_last_lr()[-1]
