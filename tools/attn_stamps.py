#!/usr/bin/env python3
"""diagnostics: cycle stamps of the persistent attention backward: python tools/attn_stamps.py [p]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_representation_learning_amd import hip
from ecg_representation_learning_amd.hip import check, ptr, stream
from toolslib import tools_lib as lib
lib().ecgvit_tools_attn_variant(2)   # the eight-wave persistent kernel of round 3 (the stamps live in it)
B, N, h, dh = 512, 251, 12, 64
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
d = h * dh; bf = torch.bfloat16
qkv = torch.randn(B * N, 3 * d, device='cuda').to(bf); out = torch.empty(B * N, d, device='cuda', dtype=bf); do = torch.randn(B * N, d, device='cuda').to(bf)
lse = torch.empty(B * h * N, device='cuda'); dqkv = torch.empty(B * N, 3 * d, device='cuda', dtype=bf)
check(lib().ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, dh, 0.125, p, 7, hip.BF16, stream()), 'f')
st = torch.zeros(768 * 128, dtype=torch.int64, device='cuda')   # one 128-word record per workgroup (the backward runs up to 768)
check(lib().ecgvit_debug_attn_stamps(ptr(st)), 'stamps')
for _ in range(3):
    check(lib().ecgvit_attention_bwd(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(dqkv), B, N, h, dh, 0.125, p, 7, hip.BF16, stream()), 'b')
torch.cuda.synchronize()
check(lib().ecgvit_debug_attn_stamps(None), 'stamps')
t = st.cpu().view(768, 4, 32).double().numpy()
for item in range(4):
    x = t[:, item]
    start, pre, post, bar, loop_end, epi_end = x[:, 0], x[:, 1:9], x[:, 9:17], x[:, 17:25], x[:, 25], x[:, 26]
    body = np.concatenate([pre[:, :1] - start[:, None], pre[:, 1:] - bar[:, :-1]], axis=1)       # start of iteration -> before the wait
    print(f'item {item}: total {np.mean(epi_end - start):.0f} cyc; loop {np.mean(loop_end - start):.0f}; drain {np.mean(epi_end - loop_end):.0f}')
    print('   compute (iteration start -> wait):', ' '.join(f'{v:.0f}' for v in body.mean(0)))
    print('   vmcnt wait                       :', ' '.join(f'{v:.0f}' for v in (post - pre).mean(0)))
    print('   barrier wait                     :', ' '.join(f'{v:.0f}' for v in (bar - post).mean(0)))
    dq = np.concatenate([pre[:, 1:] * 0, ], axis=1)
    nxt = np.concatenate([pre[:, 1:], loop_end[:, None]], axis=1)
print('gap between items (epilogue end -> next start):', np.mean(t[:, 1, 0] - t[:, 0, 26]))
