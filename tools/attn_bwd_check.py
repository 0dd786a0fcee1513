#!/usr/bin/env python3
"""where does the shipped attention backward differ from the one-item kernel (tools library)?  python tools/attn_bwd_check.py B h N p"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_representation_learning_amd import hip
from ecg_representation_learning_amd.hip import lib, check, ptr, stream
from toolslib import tools_lib
if os.environ.get('ECGVIT_AB_LIB'):
    hip.use_library(os.environ['ECGVIT_AB_LIB'])
B, h, N, p = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
dh = 64; d = h * dh; bf = torch.bfloat16
g = torch.Generator().manual_seed(5)
qkv = (torch.randn(B * N, 3 * d, generator=g) * 1.2).to(bf).cuda(); do = torch.randn(B * N, d, generator=g).to(bf).cuda()
out = torch.empty(B * N, d, device='cuda', dtype=bf); lse = torch.zeros(B * h * N, device='cuda')
check(lib().ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, dh, 0.125, p, 1234, hip.BF16, stream()), 'f')
def run(fn):
    r = torch.full((B * N, 3 * d), float('nan'), device='cuda', dtype=bf)
    check(fn(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(r), B, N, h, dh, 0.125, p, 1234, hip.BF16, stream()), 'b')
    torch.cuda.synchronize()
    return r.float().view(B, N, 3, h, dh)
new = run(lib().ecgvit_attention_bwd)
new2 = run(lib().ecgvit_attention_bwd)
print('repeatable:', torch.equal(torch.nan_to_num(new, nan=123.), torch.nan_to_num(new2, nan=123.)))
old = run(tools_lib().ecgvit_attention_bwd_oneitem) if N <= 256 else None
for i, nm in enumerate('qkv'):
    x = new[:, :, i]
    bad = ~torch.isfinite(x)
    print(f'd{nm}: non-finite {int(bad.sum())}', end='')
    if bad.any():
        idx = bad.nonzero()
        print(' at (b, n, head, dh) e.g.', idx[:6].tolist(), ' tokens:', sorted(set(idx[:, 1].tolist()))[:20], ' heads:', sorted(set(idx[:, 2].tolist())), end='')
    if old is not None:
        y = old[:, :, i]
        ok = torch.isfinite(x) & torch.isfinite(y)
        rel = float((x[ok] - y[ok]).norm() / y[ok].norm())
        err = (x - y).abs(); err[~ok] = 0
        w = err.view(B, N, -1).amax(-1)
        print(f'  rel err vs one-item {rel:.2e}; max abs {float(err.max()):.3e}; worst tokens {w.amax(0).topk(5).indices.tolist()}; worst records {w.amax(1).topk(3).indices.tolist()}', end='')
    print()
