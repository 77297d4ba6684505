"""The optimisation-step protocol of the reference's ``training/gim_img_training.py``
(im_eval_step :76, au_eval_step :85, im_train_step :157, au_train_step :169) on the MI355X engine, plus a fused
convenience ``gim_step`` (= im_train_step then au_train_step, as train_epoch does at :225-239)."""
import os

import torch

from . import ops

_OVERLAP = os.environ.get("GIM_NO_STEP_OVERLAP") is None  # A/B switch


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


def gim_step(trainer, leaked_sample, real_sample, si_sample, z=None, overlap=None):
    """One training iteration on one episode batch: generator step then discriminator step on the fake
    sample produced with the pre-update generator (training/gim_img_training.py:225-239, n_au_steps=1).

    On the GPU the two steps overlap: the discriminator step depends on the generator's FORWARD only (the fake sample),
    not on its backward or update, and the generator's backward only reads the discriminator's weights.  So the
    discriminator step runs on its own stream (lane 1) next to the generator's backward, with two orderings kept:
    the discriminator's Adam update waits for the generator's backward (which reads those weights), and the caller's
    stream waits for the discriminator step at the end.  Results are those of the sequential protocol.
    overlap=False (or GIM_NO_STEP_OVERLAP=1) runs the two steps back to back; hipGraph capture does (a hipGraph replays
    parallel branches slower than eager streams run them: 234 vs 282 episodes/s)."""
    if overlap is None:
        overlap = _OVERLAP
    if not (overlap and leaked_sample.is_cuda):
        im = im_train_step(trainer, leaked_sample, si_sample, z=z)
        au = au_train_step(trainer, real_sample, im[1], si_sample)
        return im, au
    from .gim_img_models import lane_stream
    mod = trainer.module
    cur = torch.cuda.current_stream()
    dstream = lane_stream(leaked_sample.device, 1)

    # generator: forward on the caller's stream
    mod.impersonator.train()
    mod.impersonator_opt.zero_grad()
    loss, fake_sample, au_out = trainer.forward(mode='impersonator_forward', leaked_sample=leaked_sample,
                                                si_sample=si_sample, **({} if z is None else {"z": z}))
    loss = loss.mean()
    fake_d = fake_sample.detach()
    dstream.wait_stream(cur)            # lane 1 forks here: everything up to the generator's forward is visible to it
    for t in (fake_d, real_sample, si_sample):
        t.record_stream(dstream)

    # generator: backward on the caller's stream (enqueued first: it is the longer dependency chain)
    loss.backward()
    gbwd_done = cur.record_event()
    im = (loss.detach(), fake_d, au_out.detach())

    # discriminator step on lane 1
    with torch.cuda.stream(dstream), ops.lane(1):
        mod.authenticator.train()
        mod.authenticator_opt.zero_grad()
        (dloss, loss_on_real, loss_on_fake, reg, out_on_real, out_on_fake, pred_on_real, pred_on_fake,
         fake_out) = trainer.forward(mode='authenticator_forward', fake_sample=fake_d, real_sample=real_sample,
                                     si_sample=si_sample)
        dloss = dloss.mean()
        dloss.backward()
        dstream.wait_event(gbwd_done)   # the generator's backward reads the weights this update overwrites
        mod.authenticator_opt.step()
        au = (dloss.detach(), loss_on_real.detach().mean(), loss_on_fake.detach().mean(), reg.detach().mean(),
              out_on_real.detach().mean(), out_on_fake.detach().mean(),
              pred_on_real.detach(), pred_on_fake.detach(), fake_out.detach())
    mod.impersonator_opt.step()         # generator's Adam: nothing on lane 1 reads the generator's weights
    cur.wait_stream(dstream)
    for t in au:
        t.record_stream(cur)
    return im, au
