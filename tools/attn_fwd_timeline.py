#!/usr/bin/env python3
"""diagnostics: where a workgroup of the attention forward spends its life, and how busy a CU's two workgroup slots are.
Per workgroup the tools build stamps {start, images landed, last product done, stores issued} on the 100-MHz clock plus HW_ID.
usage (GPU box): python tools/attn_fwd_timeline.py [p]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_representation_learning_amd import hip
from ecg_representation_learning_amd.hip import check, ptr, stream
from toolslib import tools_lib as lib
B, N, h, dh = 512, 251, 12, 64
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
d = h * dh; bf = torch.bfloat16
qkv = torch.randn(B * N, 3 * d, device='cuda').to(bf); out = torch.empty(B * N, d, device='cuda', dtype=bf)
lse = torch.empty(B * h * N, device='cuda')
st = torch.zeros(768 * 128, dtype=torch.int64, device='cuda')
for _ in range(3):
    check(lib().ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, dh, 0.125, p, 7, hip.BF16, stream()), 'f')
torch.cuda.synchronize()
check(lib().ecgvit_debug_attn_stamps(ptr(st)), 'stamps')
check(lib().ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, dh, 0.125, p, 7, hip.BF16, stream()), 'f')
torch.cuda.synchronize()
check(lib().ecgvit_debug_attn_stamps(None), 'stamps')
t = st.cpu().view(-1, 8)[:B * h].double().numpy()
t0 = t[:, 0].min()
s0, s1, s2, s3 = [(t[:, k] - t0) / 100.0 for k in range(4)]   # us
hw = t[:, 4].astype(np.int64)
print(f'{B * h} workgroups; kernel span (first start -> last end) {s3.max():.1f} us')
print(f'per workgroup, us (mean / p10 / p90):')
for name, v in (('start -> images landed', s1 - s0), ('images landed -> last product', s2 - s1), ('last product -> stores issued', s3 - s2), ('whole life', s3 - s0)):
    print(f'   {name:32s} {v.mean():6.2f} / {np.percentile(v, 10):6.2f} / {np.percentile(v, 90):6.2f}')
# CU identity from HW_ID (gfx9: cu_id bits 8-11, sh_id 12, se_id 13-15 ... xcc in XCC_ID; use the whole masked word as a key)
key = hw & 0xFFFFFF00 | 0
ids, inv = np.unique(hw >> 8 & 0xFFFF, return_inverse=True)
print(f'distinct (se, sh, cu) keys seen: {len(ids)} (x 8 XCDs share keys: slots per key = {B * h / len(ids):.1f} workgroups)')
order = np.argsort(s0)
print('start times of the first 16 dispatched:', ' '.join(f'{v:.1f}' for v in s0[order][:16]))
print(f'last start {s0.max():.1f} us; workgroups started per us in the steady part: {(B * h - 512) / max(1e-9, (s0.max() - np.sort(s0)[512])):.1f}')
busy = (s3 - s0).sum() / (512 * s3.max())
print(f'slot occupancy: sum of lives / (512 slots x span) = {busy:.3f}')
