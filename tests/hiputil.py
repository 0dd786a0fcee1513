"""helpers for the -m gpu tests: call the C-ABI directly with torch device tensors"""
import torch

import ecg_representation_learning_amd as E
from ecg_representation_learning_amd import hip
from ecg_representation_learning_amd.hip import lib, check, ptr, stream


def tools_lib():
    """the TOOLS build of the library (tools/ecgvit_hip_tools.h): test-only entry points -- the fragment-layout probe and the one-item attention
    backward the persistent kernel is held against.  Built by __graft_entry__.build(); never loaded by the product package."""
    import os
    import sys
    tools_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools')
    if tools_dir not in sys.path:
        sys.path.insert(0, tools_dir)
    import toolslib
    return toolslib.tools_lib()


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
