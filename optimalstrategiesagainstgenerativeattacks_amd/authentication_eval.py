"""Inference path of the reference's ``authentication_eval`` on the engine:
``eval_gim_on_authentication.py:25-44,75-106`` (model -> agent wrappers), ``agents.py:16-58`` (Authenticator / Impersonator
agents, replay impersonator) and ``authentication_score.py:32-97`` (accuracy / ROC-AUC over a dataset).  Host-side glue
over the same modules: both networks run in eval mode under ``torch.no_grad()`` (no power iteration, sigma from the stored
u, v); batches come from ``data.EpisodeBank.gpu_batches`` or a DataLoader, and per-batch statistics stay on the device."""
import itertools
import random

import numpy as np
import torch
from tqdm import tqdm

from .gim_img_training import _batches, _world


def get_au_function(au):
    def au_model_func(test_sample, si_sample):
        with torch.no_grad():
            (au_si_src, au_test_src), (au_si_env, au_test_env) = au.encode_samples([si_sample, test_sample])
            out = au.dis(test_src=au_test_src, test_env=au_test_env, si_src=au_si_src, si_env=au_si_env)
        return out.detach()
    return au_model_func


def get_im_function(im, args_dict):
    def im_model_func(leaked_sample, n):
        with torch.no_grad():
            fake_sample = im.forward(leaked_sample=leaked_sample, n=n, remove_noise_mean=args_dict['remove_noise_mean'])
        return fake_sample.detach()
    return im_model_func


class Authenticator:
    def __init__(self, au_model_func, th=0.):
        self.au_model_func = au_model_func
        self.th = th

    def act(self, test_sample, si_sample):
        out = self.au_model_func(test_sample=test_sample, si_sample=si_sample)
        pred = torch.ge(out, self.th).to(torch.long)
        return out, pred


class Impersonator:
    def __init__(self, im_model_func):
        self.im_model_func = im_model_func

    def act(self, leaked_sample, n):
        return self.im_model_func(leaked_sample=leaked_sample, n=n)


def replay_impersonator(leaked_sample, n):
    m = leaked_sample.size(1)
    return torch.cat([leaked_sample[:, random.randrange(m)].unsqueeze(1) for _ in range(n)], dim=1)


def get_gim_authenticator(au):
    """agents.Authenticator around an engine authenticator (eval mode)."""
    au.eval()
    return Authenticator(get_au_function(au))


def get_gim_impersonator(im, args_dict):
    im.eval()
    return Impersonator(get_im_function(im, args_dict))


def comp_acc(pred_on_real, pred_on_fake):
    assert len(pred_on_real.size()) == 1 and len(pred_on_fake.size()) == 1
    assert pred_on_real.size(0) == pred_on_fake.size(0)
    acc_on_real = pred_on_real.to(torch.float).mean()
    acc_on_fake = torch.eq(pred_on_fake, 0).to(torch.float).mean()
    return 0.5 * (acc_on_real + acc_on_fake), acc_on_fake, acc_on_real


def roc_auc(y_true, y_score):
    """Area under the ROC curve = Mann-Whitney U statistic with mid-ranks for ties (sklearn.metrics.roc_auc_score)."""
    y_true = np.asarray(y_true).astype(bool)
    y_score = np.asarray(y_score, dtype=np.float64)
    order = np.argsort(y_score, kind="mergesort")
    ranks = np.empty(len(y_score), dtype=np.float64)
    s = y_score[order]
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    n_pos, n_neg = int(y_true.sum()), int((~y_true).sum())
    return float((ranks[y_true].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def eval_authenticator_and_impersonator(device, ds, batch_size, num_workers, authenticator, impersonator, dbg=False):
    """authentication_score.py:45-97: accuracy (total, on fake, on real) and ROC-AUC of `authenticator` against `impersonator`."""
    outs = {"or": [], "of": [], "pr": [], "pf": []}
    batches, n_batches = _batches(ds, batch_size, True, num_workers, device)
    num_iters = min(1000, n_batches) if dbg else n_batches
    rank, _ = _world()
    for data_batch in tqdm(itertools.islice(batches, num_iters), total=num_iters, desc='Eval Authentication', disable=rank != 0):
        real_sample, leaked_sample, si_sample = data_batch["real_sample"], data_batch["leaked_sample"], data_batch["si_sample"]
        n = real_sample.size(1)
        out_on_real, pred_on_real = authenticator.act(test_sample=real_sample, si_sample=si_sample)
        fake_sample = impersonator.act(leaked_sample=leaked_sample, n=n)
        out_on_fake, pred_on_fake = authenticator.act(test_sample=fake_sample, si_sample=si_sample)
        for k_, v in (("or", out_on_real), ("of", out_on_fake), ("pr", pred_on_real), ("pf", pred_on_fake)):
            outs[k_].append(v.view(-1).detach())
    out_on_real, out_on_fake = torch.cat(outs["or"]), torch.cat(outs["of"])
    acc, acc_on_fake, acc_on_real = comp_acc(pred_on_real=torch.cat(outs["pr"]), pred_on_fake=torch.cat(outs["pf"]))
    y_true = torch.cat([torch.ones_like(out_on_real), torch.zeros_like(out_on_fake)]).cpu().numpy()
    y_score = torch.cat([out_on_real, out_on_fake]).cpu().numpy()
    return acc, acc_on_fake, acc_on_real, roc_auc(y_true, y_score)
