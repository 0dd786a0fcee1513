"""helpers for the -m gpu tests: call the C-ABI directly with torch device tensors"""
import torch

import ecg_representation_learning_amd as E
from ecg_representation_learning_amd import hip
from ecg_representation_learning_amd.hip import lib, check, ptr, stream


_keepalive = []  # device copies made by dev() stay alive until the test ends (raw pointers are handed to the C-ABI)


def dev(t, dtype=None):
    t = torch.as_tensor(t)
    if dtype is not None:
        t = t.to(dtype)
    d = t.contiguous().cuda()
    _keepalive.append(d)
    return d


def release():
    _keepalive.clear()


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max())


def gelu(x):
    return torch.nn.functional.gelu(x)


def gelu_grad(x):
    x = x.double()
    cdf = 0.5 * (1 + torch.erf(x / 2 ** 0.5))
    pdf = torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    return (cdf + x * pdf)
