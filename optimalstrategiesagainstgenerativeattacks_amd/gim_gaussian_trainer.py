"""GIMGaussianTrainer (training/gim_gaussian_trainer.py:20-150) on the engine: same constructor, forward modes and
return tuples as the reference (R1 term included); FusedAdam with torch.optim.Adam's default betas.  Everything the two
trainers have in common lives in gim_img_trainer.GimTrainerBase."""
from .gim_img_trainer import GimTrainerBase, _frozen
from .optim import FusedAdam


class GIMGaussianTrainer(GimTrainerBase):
    def __init__(self, outdir, m, n, k, authenticator, impersonator, au_lr, im_lr, reg_param=0., remove_noise_mean=True):
        super().__init__()
        self._init_agents(m, n, k, authenticator, impersonator, reg_param, remove_noise_mean)
        self.authenticator_opt = FusedAdam(self.authenticator.parameters(), lr=au_lr)
        self.impersonator_opt = FusedAdam(self.impersonator.parameters(), lr=im_lr)
        self._init_checkpoints(outdir)

    def authenticator_forward(self, fake_sample, real_sample, si_sample, grad=True):
        self._wants_input_grad(real_sample, si_sample)
        return self._discriminator_terms(self.authenticator(test_sample=real_sample, si_sample=si_sample),
                                         lambda: self.authenticator(test_sample=fake_sample, si_sample=si_sample),
                                         real_sample, si_sample, fake_sample, grad)

    def impersonator_forward(self, leaked_sample, si_sample, z=None):
        fake_sample = self.impersonator(leaked_sample=leaked_sample, n=self.n, remove_noise_mean=self.remove_noise_mean, z=z)
        with _frozen(self.authenticator):
            auth_out = self.authenticator(test_sample=fake_sample, si_sample=si_sample)
        return self.gan_loss(dis_out=auth_out, target=1.), fake_sample, auth_out

    def save(self):
        step = self.global_step
        print("\nSaving checkpoint...\n")
        self.checkpoint_io.save(global_step=step, last_epoch=1, filename="model_{:08}.pt".format(step))
