#!/usr/bin/env python3
"""Micro-benchmark of the bf16 GEMM kernels at the EcgVit-base train-step shapes (M = 512*251 tokens).
usage: python tools/gemm_bench.py [--iters 20]      (set ECGVIT_GEMM_V2=0 to time the 128^2 kernel)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecg_representation_learning_amd as E  # noqa: E402
from ecg_representation_learning_amd import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--m', type=int, default=512 * 251)
    ap.add_argument('--library', action='store_true', help='also time torch.matmul (hipBLASLt) on the same shapes: a yardstick only')
    args = ap.parse_args()
    M = args.m
    d, f = 768, 3072
    shapes = [('qkv', d, 3 * d), ('out', d, d), ('ffn_up', d, f), ('ffn_down', f, d)]
    bf = torch.bfloat16
    ws = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
    for name, kin, nout in shapes:
        X = torch.randn(M, kin, device='cuda').to(bf)
        W = (torch.randn(nout, kin, device='cuda') * 0.02).to(bf)
        dY = torch.randn(M, nout, device='cuda').to(bf)
        Y = torch.empty(M, nout, device='cuda', dtype=bf)
        dX = torch.empty(M, kin, device='cuda', dtype=bf)
        dW = torch.empty(nout, kin, device='cuda', dtype=torch.float32)
        runs = {
            'fwd NT': lambda: hip.gemm(hip.GEMM_NT, X, W, Y, M, nout, kin, kin, kin, nout),
            'dgrad NN': lambda: hip.gemm(hip.GEMM_NN, dY, W, dX, M, kin, nout, nout, kin, kin),
            'wgrad TN': lambda: hip.gemm(hip.GEMM_TN, dY, X, dW, nout, kin, M, nout, kin, kin, workspace=ws),
        }
        if args.library:
            dWl = torch.empty(nout, kin, device='cuda', dtype=bf)
            runs.update({
                'lib NT': lambda: torch.matmul(X, W.t(), out=Y),
                'lib NN': lambda: torch.matmul(dY, W, out=dX),
                'lib TN': lambda: torch.matmul(dY.t(), X, out=dWl),
            })
        for tag, fn in runs.items():
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            tf = 2.0 * M * kin * nout / (ms * 1e-3) / 1e12
            print(f'{name:9s} {tag:9s} M={M} K/N={kin}/{nout}: {ms * 1e3:8.1f} us  {tf:7.1f} TFLOP/s  ({100 * tf / 2500:4.1f} % of bf16 peak)', flush=True)


if __name__ == '__main__':
    main()
