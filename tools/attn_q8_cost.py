#!/usr/bin/env python3
"""what do the 8-bit emitting attention entry points cost over the plain ones?  One process, interleaved rounds, EcgVit-large / 501 tokens
(256 x 16 x 501) and EcgVit-base (512 x 12 x 251).  usage: python tools/attn_q8_cost.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecg_representation_learning_amd import hip  # noqa: E402
from ecg_representation_learning_amd.hip import lib, check, ptr, stream  # noqa: E402
import bench as _bench  # noqa: E402

print('kernel_source_sha16:', _bench.kernel_source_hash(), flush=True)
bf = torch.bfloat16
for B, h, N in ((256, 16, 501), (512, 12, 251)):
    d = h * 64
    g = torch.Generator().manual_seed(3)
    qkv = (torch.randn(B * N, 3 * d, generator=g)).to(bf).cuda()
    do = torch.randn(B * N, d, generator=g).to(bf).cuda()
    out = torch.empty(B * N, d, device='cuda', dtype=bf)
    out8 = torch.empty(B * N, d, device='cuda', dtype=torch.uint8)
    dqkv = torch.empty(B * N, 3 * d, device='cuda', dtype=bf)
    dqkv8 = torch.empty(B * N, 3 * d, device='cuda', dtype=torch.uint8)
    lse = torch.zeros(B * h * N, device='cuda')
    sc = torch.ones(1, device='cuda')
    am = torch.zeros(1, device='cuda')
    L = lib()
    fns = {
        'fwd': lambda: check(L.ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, 64, 0.125, 0.1, 77, hip.BF16, stream()), 'f'),
        'fwd_q8': lambda: check(L.ecgvit_attention_fwd_q8(ptr(qkv), ptr(out), ptr(lse), B, N, h, 64, 0.125, 0.1, 77, ptr(out8), ptr(sc), ptr(am), stream()), 'f8'),
        'bwd': lambda: check(L.ecgvit_attention_bwd(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(dqkv), B, N, h, 64, 0.125, 0.1, 77, hip.BF16, stream()), 'b'),
        'bwd_q8': lambda: check(L.ecgvit_attention_bwd_q8(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(dqkv), B, N, h, 64, 0.125, 0.1, 77, ptr(dqkv8), ptr(sc), ptr(am), stream()), 'b8'),
    }
    times = {k: [] for k in fns}
    for f in fns.values():
        f(); f()
    torch.cuda.synchronize()
    for _ in range(7):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) * 100)
    for k in fns:
        t = sorted(times[k])
        print(f'{B} x {h} x {N}  {k:7s}: median {t[len(t) // 2]:8.1f} us  min {t[0]:8.1f} us', flush=True)
