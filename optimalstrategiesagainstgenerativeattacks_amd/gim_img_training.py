"""The optimisation-step protocol of the reference's ``training/gim_img_training.py``
(im_eval_step :76, au_eval_step :85, im_train_step :157, au_train_step :169) on the MI355X engine, plus a fused
convenience ``gim_step`` (= im_train_step then au_train_step, as train_epoch does at :225-239)."""
import itertools
import os

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader
from tqdm import tqdm

from . import model_blocks as mb
from . import ops
from .training_logger import Logger
from .training_utils import DataParallelMock, EpisodeParallel, adjust_batch_size, get_device

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


def _backward(loss):
    """loss.backward() - on the fp16 matrix path of (loss * S), S = ops.loss_scale() (the optimizer step un-scales: FusedAdam.step
    (grad_scale=1 / S)), so that the gradients the fp16 convolution kernels round stay inside fp16's precise range."""
    s = ops.loss_scale()
    with ops.caller_thread_backward():     # (the nodes' Python on this thread: 15-25 % less host time per step, see ops)
        if s != 1.0:
            (loss * s).backward()
        else:
            loss.backward()


def im_train_step(trainer, leaked_sample, si_sample, z=None):
    trainer.module.impersonator.train()
    trainer.module.impersonator_opt.zero_grad()
    loss, fake_sample, au_out = trainer.forward(mode='impersonator_forward', leaked_sample=leaked_sample,
                                                si_sample=si_sample, **({} if z is None else {"z": z}))
    loss = loss.mean()
    _backward(loss)
    trainer.module.impersonator_opt.step(grad_scale=1.0 / ops.loss_scale())
    return loss.detach(), fake_sample.detach(), au_out.detach()


def au_train_step(trainer, real_sample, fake_sample, si_sample):
    trainer.module.authenticator.train()
    trainer.module.authenticator_opt.zero_grad()
    (loss, loss_on_real, loss_on_fake, reg, out_on_real, out_on_fake, pred_on_real, pred_on_fake,
     fake_sample) = trainer.forward(mode='authenticator_forward', fake_sample=fake_sample, real_sample=real_sample,
                                    si_sample=si_sample)
    loss = loss.mean()
    _backward(loss)
    trainer.module.authenticator_opt.step(grad_scale=1.0 / ops.loss_scale())
    return (loss.detach(), loss_on_real.detach().mean(), loss_on_fake.detach().mean(), reg.detach().mean(),
            out_on_real.detach().mean(), out_on_fake.detach().mean(),
            pred_on_real.detach(), pred_on_fake.detach(), fake_sample.detach())


def gim_step(trainer, leaked_sample, real_sample, si_sample, z=None, overlap=None, defer_join=False):
    """One training iteration on one episode batch: generator step then discriminator step on the fake
    sample produced with the pre-update generator (training/gim_img_training.py:225-239, n_au_steps=1).

    On the GPU the two steps overlap: the discriminator step depends on the generator's FORWARD only (the fake sample),
    not on its backward or update, and the generator's backward only reads the discriminator's weights.  So the
    discriminator step runs on its own stream (lane 1) next to the generator's backward, with two orderings kept:
    the discriminator's Adam update waits for the generator's backward (which reads those weights), and the caller's
    stream waits for the discriminator step at the end.  Results are those of the sequential protocol.
    defer_join=True additionally leaves the caller's stream un-joined at return, so that the generator part of the NEXT
    iteration's forward runs under the tail of this discriminator step; the join happens where the discriminator's weights
    or outputs are next needed (ops.join_lanes(): inside the trainer's forward modes, save(), and by callers before they read
    the returned discriminator statistics).
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
    ops.mark_phase("step start")
    mod.impersonator.train()
    mod.impersonator_opt.zero_grad()
    loss, fake_sample, au_out = trainer.forward(mode='impersonator_forward', leaked_sample=leaked_sample,
                                                si_sample=si_sample, **({} if z is None else {"z": z}))
    loss = loss.mean()
    fake_d = fake_sample.detach()
    ops.mark_phase("G forward done")
    ops.stream_wait(dstream, cur)            # lane 1 forks here: everything up to the generator's forward is visible to it
    for t in (fake_d, real_sample, si_sample):
        t.record_stream(dstream)

    # generator: backward on the caller's stream (enqueued first: it is the longer dependency chain; enqueueing the discriminator's
    # FORWARD ahead of it - so that lane 1 starts ~8 ms of host time earlier - was measured 4 % SLOWER, 406 vs 423 episodes/s over
    # three alternating pairs on one box: lane 1's early kernels take the chip from the critical lane; profiles/r04_d_*)
    _backward(loss)
    ops.mark_phase("G backward done")
    gbwd_done = cur.record_event()
    # generator's Adam right away: nothing on lane 1 reads the generator's weights, and with several GPUs its gradient
    # all-reduce (the larger bucket, 246 MB) then runs under the discriminator step instead of after it
    mod.impersonator_opt.step(grad_scale=1.0 / ops.loss_scale())
    ops.mark_phase("G update done")
    im = (loss.detach(), fake_d, au_out.detach())

    # discriminator step on lane 1
    with torch.cuda.stream(dstream), ops.lane(1):
        ops.mark_phase("D start")
        mod.authenticator.train()
        mod.authenticator_opt.zero_grad()
        (dloss, loss_on_real, loss_on_fake, reg, out_on_real, out_on_fake, pred_on_real, pred_on_fake,
         fake_out) = trainer.forward(mode='authenticator_forward', fake_sample=fake_d, real_sample=real_sample,
                                     si_sample=si_sample)
        dloss = dloss.mean()
        ops.mark_phase("D forward done")
        _backward(dloss)
        ops.mark_phase("D backward done")
        dstream.wait_event(gbwd_done)   # the generator's backward reads the weights this update overwrites
        mod.authenticator_opt.step(grad_scale=1.0 / ops.loss_scale())
        ops.mark_phase("D update done")
        au = (dloss.detach(), loss_on_real.detach().mean(), loss_on_fake.detach().mean(), reg.detach().mean(),
              out_on_real.detach().mean(), out_on_fake.detach().mean(),
              pred_on_real.detach(), pred_on_fake.detach(), fake_out.detach())
    if defer_join:
        ops.defer_join(dstream)
    else:
        ops.stream_wait(cur, dstream)
    for t in au:
        t.record_stream(cur)
    return im, au


# --------------------------------------------------------------------------------------------------------------------
# the caller loop (training/gim_img_training.py:24-74, 98-154, 186-354, 356-441)
# --------------------------------------------------------------------------------------------------------------------
def save_imgs(logger, img_sample, category, k, global_step):
    if logger is None:   # a rank other than 0 of a data-parallel job: it ran the forward (see train_epoch) and logs nothing
        return
    imgs_for_save = ((img_sample[0].clamp(-1, 1) + 1) / 2.0).cpu()
    logger.add_imgs(imgs=imgs_for_save, category=category, k=k, global_step=global_step)


def sample_and_save_imgs(device, logger, trainer, ds, ds_prefix, indices, dbg=False):
    with torch.no_grad():
        global_step = trainer.module.get_global_step()
        for idx in indices:
            data = ds[idx]
            leaked_sample = data["leaked_sample"].unsqueeze(0).to(device)
            fake_sample = trainer.forward(mode='impersonator_sample', leaked_sample=leaked_sample)
            cat = "{} imgs_{:04}".format(ds_prefix, idx)
            save_imgs(logger=logger, img_sample=leaked_sample, category=cat, k="leaked", global_step=global_step)
            save_imgs(logger=logger, img_sample=fake_sample, category=cat, k="impersonator", global_step=global_step)
            if dbg:
                save_imgs(logger=logger, img_sample=data["real_sample"].unsqueeze(0).to(device), category=cat, k="real", global_step=global_step)
                save_imgs(logger=logger, img_sample=data["si_sample"].unsqueeze(0).to(device), category=cat, k="si", global_step=global_step)


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def _batches(ds, batch_size, shuffle, num_workers, device):
    """(iterator of batch dicts on `device`, number of batches).  A dataset that batches on the GPU itself
    (data.EpisodeBank.gpu_batches) is used directly; anything else goes through the reference's DataLoader."""
    rank, world = _world()
    if hasattr(ds, "gpu_batches"):
        return ds.gpu_batches(batch_size, shuffle, drop_last=True, rank=rank, world=world), ds.num_batches(batch_size)
    loader = DataLoader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, drop_last=True)
    per = batch_size // world

    def it():
        for b in loader:
            yield {k_: (v[rank * per:(rank + 1) * per].to(device, non_blocking=True) if torch.is_tensor(v) else v) for k_, v in b.items()}
    return it(), len(loader)


def eval_step(device, trainer, ds, logger, batch_size):
    """training/gim_img_training.py:98-154: one pass over `ds` with both agents in eval mode; per-batch statistics stay
    on the device and are fetched once at the end."""
    keys = ("au_loss", "au_loss_on_real", "au_loss_on_fake", "au_out_on_real", "au_out_on_fake", "au_acc", "au_acc_on_real",
            "au_acc_on_fake", "im_loss")
    acc = {k_: [] for k_ in keys}
    global_step = trainer.module.get_global_step()
    batches, num_iters = _batches(ds, batch_size, False, 0, device)
    rank, _ = _world()
    for data_batch in tqdm(itertools.islice(batches, num_iters), total=num_iters, desc='Eval', disable=rank != 0):
        real_sample, leaked_sample, si_sample = data_batch["real_sample"], data_batch["leaked_sample"], data_batch["si_sample"]
        im_loss, fake_sample, _ = im_eval_step(trainer=trainer, leaked_sample=leaked_sample, si_sample=si_sample)
        (au_loss, au_loss_on_real, au_loss_on_fake, au_reg, au_out_on_real, au_out_on_fake, au_pred_on_real, au_pred_on_fake,
         fake_sample) = au_eval_step(trainer=trainer, real_sample=real_sample, fake_sample=fake_sample, si_sample=si_sample)
        au_acc_on_real = au_pred_on_real.to(torch.float).mean()
        au_acc_on_fake = torch.eq(au_pred_on_fake, 0).to(torch.float).mean()
        vals = (au_loss, au_loss_on_real, au_loss_on_fake, au_out_on_real, au_out_on_fake, 0.5 * (au_acc_on_real + au_acc_on_fake),
                au_acc_on_real, au_acc_on_fake, im_loss)
        for k_, v in zip(keys, vals):
            acc[k_].append(v.view(1))
    if not acc["au_loss"]:
        return
    means = torch.stack([torch.cat(acc[k_]).mean() for k_ in keys]).tolist()   # one device -> host transfer
    names = (('eval losses', 'dis loss'), ('eval losses', 'dis loss on real'), ('eval losses', 'dis loss on fake'),
             ('eval au out', 'au out on real'), ('eval au out', 'au out on fake'), ('eval accuracy', 'dis acc'),
             ('eval accuracy', 'dis acc on real'), ('eval accuracy', 'dis acc on fake'), ('eval losses', 'gen loss'))
    for (category, k_), v in zip(names, means):
        logger.add_scalar(category=category, k=k_, v=v, global_step=global_step)


def train_epoch(device, logger, epoch, trainer, train_ds, val_ds, train_batch_size, val_batch_size, num_workers,
                save_every, eval_every, save_imgs_every, train_eval_indices, val_eval_indices,
                tb_log_every=100, tb_log_enc_every=500, n_au_steps=1, dbg=False):
    """training/gim_img_training.py:186-354.  Same protocol and cadences; the generator step and the discriminator step of
    an iteration are issued through gim_step (overlapped on two streams) when the generator trains this iteration.  Nothing
    is fetched from the device between log points."""
    buf = {k_: [] for k_ in ("au_loss", "au_loss_on_real", "au_loss_on_fake", "au_reg", "au_out_on_real", "au_out_on_fake",
                             "au_pred_on_real", "au_pred_on_fake", "im_loss")}
    rank, _ = _world()
    batches, n_batches = _batches(train_ds, train_batch_size, True, num_workers, device)
    num_iters = min(50, n_batches) if dbg else n_batches
    for data_batch in tqdm(itertools.islice(batches, num_iters), total=num_iters, desc='Training', disable=rank != 0):
        trainer.module.do_global_step()
        trainer.module.update_learning_rate()
        real_sample, leaked_sample, si_sample = data_batch["real_sample"], data_batch["leaked_sample"], data_batch["si_sample"]
        global_step = trainer.module.global_step

        if (global_step + 1) % n_au_steps == 0:
            (im_loss, fake_sample, _), au = gim_step(trainer, leaked_sample, real_sample, si_sample, defer_join=True)
        else:
            im_loss, fake_sample, _ = im_eval_step(trainer=trainer, leaked_sample=leaked_sample, si_sample=si_sample)
            au = au_train_step(trainer=trainer, real_sample=real_sample, fake_sample=fake_sample, si_sample=si_sample)
        au_loss, au_loss_on_real, au_loss_on_fake, au_reg, au_out_on_real, au_out_on_fake, au_pred_on_real, au_pred_on_fake, fake_sample = au
        buf["im_loss"].append(im_loss.view(1))
        for k_, v in (("au_loss", au_loss), ("au_loss_on_real", au_loss_on_real), ("au_loss_on_fake", au_loss_on_fake),
                      ("au_reg", au_reg), ("au_out_on_real", au_out_on_real), ("au_out_on_fake", au_out_on_fake)):
            buf[k_].append(v.view(1))
        buf["au_pred_on_real"].append(au_pred_on_real.view(-1))
        buf["au_pred_on_fake"].append(au_pred_on_fake.view(-1))

        if global_step % tb_log_every == 0:
            ops.join_lanes()
            logger.add_scalar(category='lr', k='au', v=trainer.module.au_lr, global_step=global_step)
            logger.add_scalar(category='lr', k='im', v=trainer.module.im_lr, global_step=global_step)
            logger.add_scalar(category='lr', k='im_lm', v=trainer.module.im_noise_mapping_lr, global_step=global_step)
            acc_real = torch.cat(buf["au_pred_on_real"]).to(torch.float).mean()
            acc_fake = torch.eq(torch.cat(buf["au_pred_on_fake"]), 0).to(torch.float).mean()
            vals = [torch.cat(buf[k_]).mean() for k_ in ("au_loss", "au_loss_on_real", "au_loss_on_fake", "au_reg", "au_out_on_real",
                                                         "au_out_on_fake")] + [0.5 * (acc_real + acc_fake), acc_real, acc_fake,
                                                                               torch.cat(buf["im_loss"]).mean()]
            vals = torch.stack(vals).tolist()   # the only device -> host transfer of the log point
            names = (('train_losses', 'dis_loss'), ('train_losses', 'dis_loss_on_real'), ('train_losses', 'dis_loss_on_fake'),
                     ('train_losses', 'dis_reg'), ('train_au_out', 'au_out_on_real'), ('train_au_out', 'au_out_on_fake'),
                     ('train_accuracy', 'dis_acc'), ('train_accuracy', 'dis_acc_on_real'), ('train_accuracy', 'dis_acc_on_fake'),
                     ('train losses', 'gen loss'))
            for (category, k_), v in zip(names, vals):
                logger.add_scalar(category=category, k=k_, v=v, global_step=global_step)
            for k_ in buf:
                buf[k_] = []

        if global_step % tb_log_enc_every == 0:
            ops.join_lanes()
            with torch.no_grad():
                au = trainer.module.authenticator
                enc = {}
                for nm, smp in (("real", real_sample), ("si", si_sample), ("fake", fake_sample)):
                    enc[nm + "_src"] = au.src_encode_sample(smp)
                    enc[nm + "_env"] = au.env_encode_sample(smp)
                for kind in ("src", "env"):
                    for nm in ("real", "fake"):
                        d = torch.abs(enc[nm + "_" + kind].mean(1) - enc["si_" + kind].mean(1)).mean().item()
                        logger.add_scalar(category='train-au_%s_mean' % kind, k='abs[%s-si]' % nm, v=d, global_step=global_step)
                    for nm in ("real", "si", "fake"):
                        logger.add_scalar(category='train-au_%s_std' % kind, k=nm, v=mb.custom_std(enc[nm + "_" + kind]).mean().item(),
                                          global_step=global_step)

        if global_step % save_every == 0 and rank == 0:
            trainer.module.save(epoch=epoch)
        if global_step % save_imgs_every == 0:
            # EVERY rank runs these generator forwards (only rank 0 writes the images): the generator is in train mode here, as in
            # the reference, so each forward runs one spectral-norm power iteration per conv - a rank that skipped them would
            # carry different u / v buffers, hence different sigma, from then on
            lg = logger if rank == 0 else None
            sample_and_save_imgs(device=device, logger=lg, trainer=trainer, ds=train_ds, ds_prefix='train', indices=train_eval_indices, dbg=dbg)
            sample_and_save_imgs(device=device, logger=lg, trainer=trainer, ds=val_ds, ds_prefix='val', indices=val_eval_indices, dbg=dbg)
        if global_step % eval_every == 0:
            eval_step(device=device, trainer=trainer, ds=val_ds, logger=logger, batch_size=val_batch_size)
    ops.join_lanes()


def train_gim_imgs(device_name, device_ids, outdir, train_ds, val_ds, authenticator, impersonator, m, n, k,
                   reg_param, remove_noise_mean, au_lr, im_lr, beta1, beta2, env_noise_mapping_lr, lr_gamma, milestones,
                   resume_from_ckpt, n_epochs, batch_size, num_workers, save_every, eval_every, save_imgs_every,
                   train_eval_indices, val_eval_indices, n_au_steps=1, dbg=False):
    """training/gim_img_training.py:356-441.  `device_ids` keeps its meaning as "the GPUs of the job": with more than one,
    launch one process per GPU (torchrun) - each process builds the same trainer, takes its slice of every global batch
    and the two optimizers all-reduce their gradient buckets over RCCL (EpisodeParallel replaces nn.DataParallel)."""
    from .gim_img_trainer import GIMImgTrainer
    from .gim_img_models import stream_concurrency_check
    from .training_utils import pin_rank_to_cores
    if dist.is_initialized() and not torch.cuda.is_initialized():
        pin_rank_to_cores()   # a core set per rank, before the first GPU call (the runtime's helper threads inherit it)
    device = get_device(device_type=device_name, device_ids=device_ids)
    stream_concurrency_check(device)   # warns when the engine's streams share a hardware queue (GPU_MAX_HW_QUEUES not in force)
    rank, world = _world()
    n_devices = world if dist.is_initialized() else 1
    assert batch_size % n_devices == 0

    logger = Logger(log_dir=os.path.join(outdir, 'logs'), img_dir=os.path.join(outdir, 'imgs'), tensorboard_dir=os.path.join(outdir, 'tb'))
    authenticator = authenticator.to(device)
    impersonator = impersonator.to(device)
    trainer = GIMImgTrainer(outdir=outdir, m=m, n=n, k=k, authenticator=authenticator, impersonator=impersonator,
                            au_lr=au_lr, im_lr=im_lr, env_noise_mapping_lr=env_noise_mapping_lr, beta1=beta1, beta2=beta2,
                            lr_milestones=milestones, lr_gamma=lr_gamma, reg_param=reg_param,
                            remove_noise_mean=remove_noise_mean).to(device)
    if resume_from_ckpt:
        trainer.resume_from_ckpt(ckpt_path=resume_from_ckpt)
        trainer.to(device)
    if n_devices > 1:
        trainer = EpisodeParallel(trainer)
        trainer.broadcast_parameters()
    else:
        trainer = DataParallelMock(trainer)

    for ep in tqdm(range(n_epochs), "Epochs", disable=rank != 0):
        try:
            train_epoch(device=device, logger=logger, epoch=ep, trainer=trainer, train_ds=train_ds, val_ds=val_ds,
                        train_batch_size=adjust_batch_size(len(train_ds), batch_size, n_devices),
                        val_batch_size=adjust_batch_size(len(val_ds), batch_size, n_devices),
                        num_workers=num_workers, save_every=save_every, eval_every=eval_every, save_imgs_every=save_imgs_every,
                        train_eval_indices=train_eval_indices, val_eval_indices=val_eval_indices, n_au_steps=n_au_steps, dbg=dbg)
        except KeyboardInterrupt:
            print("\nKeyboardInterrupt\nSaving checkpoint...\n")
            if rank == 0:
                trainer.module.save(ep)
            break
        except PermissionError as pe:
            print("\nPermissionError\n%s\nSaving checkpoint...\n" % pe)
            if rank == 0:
                trainer.module.save(ep)
            continue
    return trainer, logger
