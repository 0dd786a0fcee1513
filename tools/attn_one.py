#!/usr/bin/env python3
"""run the fused bf16 attention fwd+bwd at the base train-step shape a few times and time them (also for rocprofv3 --pmc)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_representation_learning_amd import hip
from ecg_representation_learning_amd.hip import lib, check, ptr, stream
B, N, h, dh = 512, 251, 12, 64
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = h * dh
bf = torch.bfloat16
qkv = torch.randn(B * N, 3 * d, device='cuda').to(bf)
out = torch.empty(B * N, d, device='cuda', dtype=bf); do = torch.randn(B * N, d, device='cuda').to(bf)
lse = torch.empty(B * h * N, device='cuda'); dqkv = torch.empty(B * N, 3 * d, device='cuda', dtype=bf)
def fwd(): check(lib().ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, dh, 0.125, p, 7, hip.BF16, stream()), 'f')
def bwd(): check(lib().ecgvit_attention_bwd(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(dqkv), B, N, h, dh, 0.125, p, 7, hip.BF16, stream()), 'b')
for name, fn, fl in (('fwd', fwd, 4), ('bwd', bwd, 10)):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f'attention {name} p={p}: {us:8.1f} us  {fl * B * h * N * N * dh / us / 1e6:7.1f} TFLOP/s', flush=True)
