#!/usr/bin/env python3
"""A/B of the persistent attention backward's stagger / static-priority variants (tools build), interleaved in one process on one
device at the benchmark shape: python tools/attn_variants.py"""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecg_representation_learning_amd import hip
from ecg_representation_learning_amd.hip import check, ptr, stream
from toolslib import tools_lib as lib
import bench as _bench  # noqa: E402
print('kernel_source_sha16:', _bench.kernel_source_hash(), '(sources of the library build measured: tools/check_profiles.py holds committed tables to the round\'s bench line)', flush=True)
NAMES = {-2: "four-wave one-wave-per-SIMD experiment", 0: 'lockstep, waves 4-7 raised (round 2)', 1: 'lockstep, no priority', 2: 'staggered, waves 4-7 raised', 3: 'staggered, no priority',
         4: 'staggered, waves 0-3 raised', 5: 'lockstep, waves 0-3 raised'}
B, N, h, dh = 512, 251, 12, 64
d = h * dh; bf = torch.bfloat16
qkv = torch.randn(B * N, 3 * d, device='cuda').to(bf); out = torch.empty(B * N, d, device='cuda', dtype=bf); do = torch.randn(B * N, d, device='cuda').to(bf)
lse = torch.empty(B * h * N, device='cuda')
l = lib()
check(l.ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, dh, 0.125, 0.1, 7, hip.BF16, stream()), 'f')
outs, times = {}, {v: [] for v in NAMES}
for rnd in range(6):
    for v in NAMES:
        l.ecgvit_tools_attn_variant(v)
        dqkv = torch.zeros(B * N, 3 * d, device='cuda', dtype=bf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            check(l.ecgvit_attention_bwd(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(dqkv), B, N, h, dh, 0.125, 0.1, 7, hip.BF16, stream()), 'b')
        e1.record(); torch.cuda.synchronize()
        times[v].append(100 * e0.elapsed_time(e1))
        if v in outs:
            assert torch.equal(outs[v], dqkv)
        outs[v] = dqkv
for v in NAMES:
    t = sorted(times[v][1:])
    print(f'variant {v} ({NAMES[v]:38s}): median {t[len(t) // 2]:6.1f} us  min {t[0]:6.1f} us   bit-identical to variant 2: {torch.equal(outs[v], outs[2])}')
