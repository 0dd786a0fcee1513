#!/usr/bin/env python3
"""diagnostics: per-tile cycle stamps of the quadrant-phased forward GEMM (run with ECGVIT_GEMM_ABLATE=4): python tools/q_stamps.py kin nout"""
import os, sys, torch
os.environ['ECGVIT_GEMM_ABLATE'] = '4'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_representation_learning_amd import hip
kin, nout = int(sys.argv[1]), int(sys.argv[2])
M, bf = 512 * 251, torch.bfloat16
X = torch.randn(M, kin, device='cuda').to(bf); W = (torch.randn(nout, kin, device='cuda') * 0.02).to(bf)
Y = torch.empty(M, nout, device='cuda', dtype=bf)
ws = torch.zeros(256 * 2 * 64 * 3, dtype=torch.int64, device='cuda')
for _ in range(3):
    hip.gemm(hip.GEMM_NT, X, W, Y, M, nout, kin, kin, kin, nout, workspace=ws)
torch.cuda.synchronize()
t = ws.cpu().view(256, 2, 64, 3).double()
ntile = ((M + 255) // 256) * ((nout + 255) // 256)
per = [ntile // 256 + (1 if b < ntile % 256 else 0) for b in range(256)]
import numpy as np
for g in (0, 1):
    main, epi, gap, tot = [], [], [], []
    for b in range(256):
        n = min(per[b], 64)
        s = t[b, g, :n]
        main += (s[:, 1] - s[:, 0]).tolist(); epi += (s[:, 2] - s[:, 1]).tolist()
        gap += (s[1:, 0] - s[:-1, 2]).tolist(); tot.append(float(s[n - 1, 2] - s[0, 0]))
    nk = kin // 64
    print(f'group {g}: main loop {np.mean(main):.0f} cyc/tile ({np.mean(main) / nk:.0f} per K-tile, min {np.min(main) / nk:.0f}, p90 {np.percentile(main, 90) / nk:.0f}); '
          f'epilogue {np.mean(epi):.0f} (min {np.min(epi):.0f}, p90 {np.percentile(epi, 90):.0f}); gap {np.mean(gap):.0f}; block total {np.mean(tot):.0f} (max {np.max(tot):.0f})')
first = t[:, 0, 0, 0]
print('block start spread (cycles):', float(first.max() - first.min()))

