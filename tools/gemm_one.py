#!/usr/bin/env python3
"""run ONE gemm shape a few times (for rocprofv3 --pmc passes): python tools/gemm_one.py NT|NN|TN kin nout [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_representation_learning_amd import hip
layout, kin, nout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
M, bf = 512 * 251, torch.bfloat16
X = torch.randn(M, kin, device='cuda').to(bf); W = (torch.randn(nout, kin, device='cuda') * 0.02).to(bf)
dY = torch.randn(M, nout, device='cuda').to(bf); Y = torch.empty(M, nout, device='cuda', dtype=bf)
dX = torch.empty(M, kin, device='cuda', dtype=bf); dW = torch.empty(nout, kin, device='cuda'); ws = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
for _ in range(iters):
    if layout == 'NT': hip.gemm(hip.GEMM_NT, X, W, Y, M, nout, kin, kin, kin, nout)
    elif layout == 'NN': hip.gemm(hip.GEMM_NN, dY, W, dX, M, kin, nout, nout, kin, kin)
    else: hip.gemm(hip.GEMM_TN, dY, X, dW, nout, kin, M, nout, kin, kin, workspace=ws)
torch.cuda.synchronize()
