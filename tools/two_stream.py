#!/usr/bin/env python3
"""experiment: do an A.B^T stream (Q kernel) and an A^T.B stream (TQ kernel) run faster concurrently on two HIP streams than back to back?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_representation_learning_amd import hip
M, d, f, bf = 512 * 251, 768, 3072, torch.bfloat16
X = torch.randn(M, d, device='cuda').to(bf); W1 = (torch.randn(f, d, device='cuda') * 0.02).to(bf); H = torch.empty(M, f, device='cuda', dtype=bf)
dY = torch.randn(M, f, device='cuda').to(bf); dW = torch.empty(f, d, device='cuda'); ws = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
W2 = (torch.randn(d, f, device='cuda') * 0.02).to(bf); Y = torch.empty(M, d, device='cuda', dtype=bf)
def nt(): hip.gemm(hip.GEMM_NT, X, W1, H, M, f, d, d, d, f); hip.gemm(hip.GEMM_NT, H, W2, Y, M, d, f, f, f, d)
def tn(): hip.gemm(hip.GEMM_TN, dY, X, dW, f, d, M, f, d, d, workspace=ws)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(concurrent, n=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if concurrent:
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1):
            for _ in range(n): nt()
        with torch.cuda.stream(s2):
            for _ in range(2 * n): tn()
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    else:
        for _ in range(n): nt(); tn(); tn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for _ in range(2): run(False, 3); run(True, 3)
for _ in range(3):
    print(f'sequential {run(False):.3f} ms   concurrent {run(True):.3f} ms  per (2 A.B^T + 2 A^T.B)', flush=True)
