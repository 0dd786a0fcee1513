#!/usr/bin/env python3
"""diagnostics (tools build): per-PHASE cycle stamps of the persistent attention backward, for lane 0 of wave 0 (leading group) and
wave 4 (trailing group, half a block late), second item of every workgroup: python tools/attn_phase_stamps.py [p]
Needs the tools library built WITH the phase stamps (they cost the schedule 13 %, so the default tools build leaves them out):
    touch ecg-representation-learning_amd/csrc/attention.hip && make -C ecg-representation-learning_amd/csrc tools TOOLS_EXTRA=-DECGVIT_ATTN_PHASE_STAMPS"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
st = torch.zeros(2 * 768 * 128, dtype=torch.int64, device='cuda')
check(lib().ecgvit_debug_attn_stamps(ptr(st)), 'stamps')
for _ in range(3):
    check(lib().ecgvit_attention_bwd(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(dqkv), B, N, h, dh, 0.125, p, 7, hip.BF16, stream()), 'b')
torch.cuda.synchronize()
check(lib().ecgvit_debug_attn_stamps(None), 'stamps')
t = st.cpu().view(2, 768, 2, 8, 8)[1].double().numpy()      # [block][group][qb][phase]
names = ['start', 'issue', 'A', 'V', 'B+W', 'C', 'wait', 'barrier']
for grp, gname in ((0, 'leading (wave 0): order issue A V B W | C'), (1, 'trailing (wave 4): order issue V W B C(j-1) A(j+1) |')):
    print(gname)
    for qb in range(2, 6):
        x = t[:, grp, qb]
        ok = x[:, 0] > 0
        x = x[ok]
        # phase end stamps in program order of the group
        order = [0, 1, 2, 3, 4, 7] if grp == 0 else [0, 1, 3, 4, 5, 2, 7]
        seq = x[:, order]
        dif = np.diff(seq, axis=1).mean(0)
        print(f'  qb {qb}: ' + '  '.join(f'{names[order[i + 1]]}:{dif[i]:.0f}' for i in range(len(dif))) + f'   | start->barrier {np.mean(seq[:, -1] - seq[:, 0]):.0f}')
    # block period = barrier(qb) - barrier(qb-1)
    per = np.mean(t[:, grp, 3, 7] - t[:, grp, 2, 7])
    print(f'  period between barriers: {per:.0f} cycles')
