#!/usr/bin/env python3
"""diagnostics (tools build): cycle stamps of the four-wave attention backward, second item of every workgroup: python tools/attn_bwd4_stamps.py [p] [N]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_representation_learning_amd import hip
from ecg_representation_learning_amd.hip import check, ptr, stream
from toolslib import tools_lib as lib
import bench as _bench  # noqa: E402
print('kernel_source_sha16:', _bench.kernel_source_hash(), '(sources of the library build measured: tools/check_profiles.py holds committed tables to the round\'s bench line)', flush=True)
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 251
B, h, dh = 512, 12, 64
d = h * dh; bf = torch.bfloat16
qkv = torch.randn(B * N, 3 * d, device='cuda').to(bf); out = torch.empty(B * N, d, device='cuda', dtype=bf); do = torch.randn(B * N, d, device='cuda').to(bf)
lse = torch.empty(B * h * N, device='cuda'); dqkv = torch.empty(B * N, 3 * d, device='cuda', dtype=bf)
check(lib().ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, dh, 0.125, p, 7, hip.BF16, stream()), 'f')
st = torch.zeros(768 * 4 * 128, dtype=torch.int64, device='cuda')
check(lib().ecgvit_tools_bwd4_stamps(ptr(st)), 'stamps')
for _ in range(3):
    check(lib().ecgvit_attention_bwd(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(dqkv), B, N, h, dh, 0.125, p, 7, hip.BF16, stream()), 'b')
torch.cuda.synchronize()
check(lib().ecgvit_tools_bwd4_stamps(None), 'stamps')
t = st.cpu().view(768, 4, 16, 8).double().numpy()
nqb = (N + 31) // 32
it = t[:, :, :nqb, :7]
names = ['feed stream + delta + lse/delta reads', 'phase 0 (tile 0 vector work)', 'phase 1 (tile 1 vector work)', 'dQ store', 'counted wait', 'barrier']
print(f'N = {N}, p = {p}: median over workgroups and waves, cycles per query block (iterations 2..{nqb - 1})')
for k, nm in enumerate(names):
    dlt = it[:, :, 2:, k + 1] - it[:, :, 2:, k]
    print(f'   {nm:42s} {np.median(dlt):8.0f}   (p10 {np.percentile(dlt, 10):.0f}, p90 {np.percentile(dlt, 90):.0f})')
blk = it[:, :, 3:, 0] - it[:, :, 2:-1, 0]
print(f'   whole iteration (top to top)                {np.median(blk):8.0f}')
ev = t[:, :, :6, 7]
names2 = ['item start -> first S / dP done', 'loop', 'tail (last dV / dK, dQ)', 'flush dK / dV']
for k, nm in enumerate(names2):
    dlt = ev[:, :, k + 1] - ev[:, :, k]
    print(f'   {nm:42s} {np.median(dlt):8.0f}')
print(f'   item total (start -> flush done)            {np.median(ev[:, :, 4] - ev[:, :, 0]):8.0f}')
