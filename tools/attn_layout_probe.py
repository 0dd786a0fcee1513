#!/usr/bin/env python3
"""does the (token, head) interleaving of qkv cost the fused attention kernels time?  The same 6144 (record, head) items once as 512 records x 12 heads
(128-B segments at a 4.6-KB pitch: what the QKV product writes) and once as 6144 records x 1 head (every item one contiguous 96-KB block: what a
head-major layout would give them).  Same kernels, same arithmetic, same bytes.  usage: python tools/attn_layout_probe.py [N] [p]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_representation_learning_amd import hip
from ecg_representation_learning_amd.hip import lib, check, ptr, stream
import bench as _bench
print('kernel_source_sha16:', _bench.kernel_source_hash())
N = int(sys.argv[1]) if len(sys.argv) > 1 else 251
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
bf = torch.bfloat16
torch.manual_seed(3)
for B, h in ((512, 12), (512 * 12, 1)):
    d = h * 64
    qkv = torch.randn(B * N, 3 * d, device='cuda').to(bf); out = torch.empty(B * N, d, device='cuda', dtype=bf); do = torch.randn(B * N, d, device='cuda').to(bf)
    lse = torch.empty(B * h * N, device='cuda'); dqkv = torch.empty(B * N, 3 * d, device='cuda', dtype=bf)
    res = {}
    for rnd in range(5):
        for k in ('fwd', 'bwd'):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                if k == 'fwd':
                    check(lib().ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, 64, 0.125, p, 7, hip.BF16, stream()), 'f')
                else:
                    check(lib().ecgvit_attention_bwd(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(dqkv), B, N, h, 64, 0.125, p, 7, hip.BF16, stream()), 'b')
            e1.record(); torch.cuda.synchronize()
            res.setdefault(k, []).append(1e2 * e0.elapsed_time(e1))
    print(f'{B} records x {h} heads x {N} tokens: forward median {sorted(res["fwd"])[2]:.1f} us, backward {sorted(res["bwd"])[2]:.1f} us')
