"""Shared test helpers: portable-fill state dicts / episodes as torch tensors."""
import json
import os

import numpy as np
import torch

from oracle import portable_fill as pf

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def T(a, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def load_keys(cfg):
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        return json.load(f)[cfg]


def filled_sd(keys_and_shapes, tag, dtype=torch.float64):
    """Ordered {key: tensor} conditioned fill (see oracle/portable_fill.py)."""
    return {k: T(pf.fill_value(k, tuple(s), tag), dtype) for k, s in keys_and_shapes}


def episode(tag, B, m, n, k, c, s, d, dtype=torch.float64):
    def img(name, t):
        return T(np.clip(pf.normal("%s/%s" % (tag, name), (B, t, c, s, s)) * 0.5, -1, 1), dtype)
    return img("leaked", m), img("real", n), img("si", k), T(pf.normal(tag + "/z", (B, n, d)), dtype)


def relerr(a, b, atol=1e-12):
    """||a-b|| / (||b|| + atol*sqrt(numel)): relative L2 error with an absolute floor so that
    quantities that are mathematically zero (e.g. a conv bias grad in front of AdaIN) compare equal."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + atol * max(b.numel(), 1) ** 0.5))


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def relerr_floor(a, b, floor):
    """||a-b|| / (||b|| + floor): for families of gradients where some members are mathematically zero
    (conv bias in front of a norm layer, attention f-bias): `floor` is set from the family's largest norm."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + floor))
