#!/usr/bin/env python3
"""the fused attention kernels alone at the benchmark shape (profiling target: rocprofv3 ... -- python3 tools/attn_only.py [reps] [N] [p])"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_representation_learning_amd import hip
from ecg_representation_learning_amd.hip import lib, check, ptr, stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 251
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
B, h, dh = (512, 12, 64) if N <= 256 else (256, 16, 64)
d = h * dh; bf = torch.bfloat16
torch.manual_seed(3)
qkv = torch.randn(B * N, 3 * d, device='cuda').to(bf); out = torch.empty(B * N, d, device='cuda', dtype=bf); do = torch.randn(B * N, d, device='cuda').to(bf)
lse = torch.empty(B * h * N, device='cuda'); dqkv = torch.empty(B * N, 3 * d, device='cuda', dtype=bf)
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for r in range(reps + 2):
    if r == 2: e[0].record()
    check(lib().ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, dh, 0.125, p, 7, hip.BF16, stream()), 'f')
tf = None
e[1].record()
for r in range(reps):
    check(lib().ecgvit_attention_bwd(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(dqkv), B, N, h, dh, 0.125, p, 7, hip.BF16, stream()), 'b')
e[2].record(); torch.cuda.synchronize()
print(f'attention {B} x {h} x {N}, p = {p}: forward {1e3 * e[0].elapsed_time(e[1]) / reps:.1f} us, backward {1e3 * e[1].elapsed_time(e[2]) / reps:.1f} us per launch')
