"""The optimisation-step protocol of the reference's ``training/gim_img_training.py``
(im_eval_step :76, au_eval_step :85, im_train_step :157, au_train_step :169) on the MI355X engine, plus a fused
convenience ``gim_step`` (= im_train_step then au_train_step, as train_epoch does at :225-239)."""
import torch


def im_eval_step(trainer, leaked_sample, si_sample, z=None):
    trainer.module.impersonator.eval()
    with torch.no_grad():
        loss, fake_sample, au_out = trainer.forward(mode='impersonator_forward', leaked_sample=leaked_sample,
                                                    si_sample=si_sample, **({} if z is None else {"z": z}))
        loss = loss.mean()
    return loss.detach(), fake_sample.detach(), au_out.detach()


def au_eval_step(trainer, real_sample, fake_sample, si_sample):
    trainer.module.authenticator.eval()
    with torch.no_grad():
        (loss, loss_on_real, loss_on_fake, reg, out_on_real, out_on_fake, pred_on_real, pred_on_fake,
         fake_sample) = trainer.forward(mode='authenticator_forward', fake_sample=fake_sample, real_sample=real_sample,
                                        si_sample=si_sample, grad=False)
        loss = loss.mean()
    return (loss.detach(), loss_on_real.detach().mean(), loss_on_fake.detach().mean(), reg.detach().mean(),
            out_on_real.detach().mean(), out_on_fake.detach().mean(),
            pred_on_real.detach(), pred_on_fake.detach(), fake_sample.detach())


def im_train_step(trainer, leaked_sample, si_sample, z=None):
    trainer.module.impersonator.train()
    trainer.module.impersonator_opt.zero_grad()
    loss, fake_sample, au_out = trainer.forward(mode='impersonator_forward', leaked_sample=leaked_sample,
                                                si_sample=si_sample, **({} if z is None else {"z": z}))
    loss = loss.mean()
    loss.backward()
    trainer.module.impersonator_opt.step()
    return loss.detach(), fake_sample.detach(), au_out.detach()


def au_train_step(trainer, real_sample, fake_sample, si_sample):
    trainer.module.authenticator.train()
    trainer.module.authenticator_opt.zero_grad()
    (loss, loss_on_real, loss_on_fake, reg, out_on_real, out_on_fake, pred_on_real, pred_on_fake,
     fake_sample) = trainer.forward(mode='authenticator_forward', fake_sample=fake_sample, real_sample=real_sample,
                                    si_sample=si_sample)
    loss = loss.mean()
    loss.backward()
    trainer.module.authenticator_opt.step()
    return (loss.detach(), loss_on_real.detach().mean(), loss_on_fake.detach().mean(), reg.detach().mean(),
            out_on_real.detach().mean(), out_on_fake.detach().mean(),
            pred_on_real.detach(), pred_on_fake.detach(), fake_sample.detach())


def gim_step(trainer, leaked_sample, real_sample, si_sample, z=None):
    """One training iteration on one episode batch: generator step then discriminator step on the fake
    sample produced with the pre-update generator (training/gim_img_training.py:225-239, n_au_steps=1)."""
    im = im_train_step(trainer, leaked_sample, si_sample, z=z)
    au = au_train_step(trainer, real_sample, im[1], si_sample)
    return im, au
