#!/usr/bin/env python3
"""bf16 vs 8-bit operands on gemm_nt_kernel at the EcgVit-large Linear shapes (M = 256 records x 501 tokens), one process, interleaved."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ecg_representation_learning_amd as E  # noqa: E402,F401
from ecg_representation_learning_amd import hip  # noqa: E402


def main():
    M, d, f = 256 * 501, 1024, 4096
    bf = torch.bfloat16
    one = torch.ones(1, device='cuda')
    for name, K, N in (('qkv', d, 3 * d), ('out', d, d), ('ffn_up', d, f), ('ffn_down', f, d)):
        X = torch.randn(M, K, device='cuda').to(bf)
        W = (torch.randn(N, K, device='cuda') * 0.03).to(bf)
        X8 = torch.randn(M, K, device='cuda').to(torch.float8_e4m3fn).view(torch.uint8)
        W8 = (torch.randn(N, K, device='cuda') * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
        C = torch.empty(M, N, device='cuda', dtype=bf)
        fns = {'bf16': lambda: hip.gemm(hip.GEMM_NT, X, W, C, M, N, K, K, K, N),
               'fp8': lambda: hip.gemm(hip.GEMM_NT, X8, W8, C, M, N, K, K, K, N, fp8_format=hip.FP8_E4M3, scale_a=one, scale_b=one)}
        t = {k: [] for k in fns}
        for k, fn in fns.items():
            fn(); fn()
        for _ in range(4):
            for k, fn in fns.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                t[k].append(e0.elapsed_time(e1) / 5 * 1e3)
        fl = 2.0 * M * N * K
        print(f'{name:9s} K={K:4d} N={N:4d}: ' + '  '.join(f'{k} {sorted(v)[len(v) // 2]:7.1f} us ({fl / sorted(v)[len(v) // 2] / 1e6:6.0f} TFLOP/s)' for k, v in t.items()), flush=True)


if __name__ == '__main__':
    main()
