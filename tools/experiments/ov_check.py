#!/usr/bin/env python3
"""gemm_ov_kernel (four waves, epilogue deferred into the next tile's main loop) against the eight-wave body and an f64 reference, FFN-up forward.
usage: python tools/ov_check.py [--m 128512] [--dim 768] [--time]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ecg_representation_learning_amd as E  # noqa: E402,F401
from ecg_representation_learning_amd import hip  # noqa: E402
hip.use_library(os.path.join(ROOT, 'ecg-representation-learning_amd', 'csrc', 'build', 'libecgvit_hip_tools.so'))
from ecg_representation_learning_amd.hip import EPI_BIAS, EPI_GELU, EPI_DROPOUT, EPI_GELU_GRAD_AUX, GEMM_NT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--m', type=int, default=512 * 251)
    ap.add_argument('--dim', type=int, default=768)
    ap.add_argument('--time', action='store_true')
    ap.add_argument('--iters', type=int, default=10)
    args = ap.parse_args()
    lib = hip.lib()
    tg = lib.ecgvit_tools_gemm
    tg.restype, tg.argtypes = ctypes.c_int, [ctypes.POINTER(hip.GemmDesc), ctypes.c_void_p] + [ctypes.c_int] * 3
    M, K, N = args.m, args.dim, 4 * args.dim
    bf = torch.bfloat16
    torch.manual_seed(1)
    X = torch.randn(M, K, device='cuda').to(bf)
    W = (torch.randn(N, K, device='cuda') * 0.05).to(bf)
    bias = torch.randn(N, device='cuda') * 0.3
    st = torch.cuda.current_stream().cuda_stream
    ok = True
    for p in (0.0, 0.1):
        epi = EPI_BIAS | EPI_GELU | EPI_GELU_GRAD_AUX | (EPI_DROPOUT if p > 0 else 0)
        outs = {}
        for name, kern, diag in (('8w', 2, 128), ('ov', 4, 0)):
            C = torch.full((M, N), float('nan'), device='cuda', dtype=bf)
            A = torch.full((M, N), float('nan'), device='cuda', dtype=bf)
            d = hip.gemm_desc(GEMM_NT, X, W, C, M, N, K, K, K, N, epilogue=epi, bias=bias, aux=A, ldaux=N, dropout_p=p, seed=1234)
            rc = tg(ctypes.byref(d), st, kern, 0, diag)
            assert rc == 0, (name, rc)
            torch.cuda.synchronize()
            outs[name] = (C, A, d)
        rows = torch.randint(0, M, (2048,), device='cuda')
        rows[:4] = torch.tensor([0, 1, M - 1, M - 2], device='cuda')
        h = X[rows].double() @ W.double().t() + bias.double()
        y = 0.5 * h * (1 + torch.erf(h / 2 ** 0.5))
        dy = 0.5 * (1 + torch.erf(h / 2 ** 0.5)) + h * torch.exp(-0.5 * h * h) / (2 * torch.pi) ** 0.5
        for name in ('8w', 'ov'):
            C, A, _ = outs[name]
            assert torch.isfinite(C.float()).all() and torch.isfinite(A.float()).all(), name + ': non-finite or unwritten output'
            c, a = C[rows].double(), A[rows].double()
            if p == 0:
                ey, ed = (c - y).abs(), (a - dy).abs()
                tol_y, tol_d = 2.0 ** -8 * y.abs() + 3e-4, 2.0 ** -8 * dy.abs() + 3e-4   # one bf16 ulp of the value + the f16 arithmetic's absolute floor
                print(f'p=0 {name}: y max abs err {ey.max():.3e} (rel to 1 ulp bound {float((ey / tol_y).max()):.2f}), dy {ed.max():.3e} ({float((ed / tol_d).max()):.2f}); mean abs y err {ey.mean():.3e} dy {ed.mean():.3e}')
                if name == 'ov':
                    ok &= bool((ey <= tol_y).all()) and bool((ed <= tol_d).all())
            else:
                keep = c != 0
                rate = 1 - keep.double().mean().item()
                same = ((c == 0) == (a == 0)).double().mean().item()     # (dy = 0 only where dropped: |dy| > 1e-3 almost everywhere)
                inv = 1 / (1 - (round(p * 128) / 128 if name == 'ov' else p))
                ey = ((c - y * inv).abs() * keep)
                tol = 2.0 ** -8 * (y * inv).abs() + 4e-4
                print(f'p={p} {name}: drop rate {rate:.4f}, zero pattern agreement hact/aux {same:.6f}, kept y max err {ey.max():.3e} ({float((ey / tol).max()):.2f} of bound)')
                if name == 'ov':
                    ok &= abs(rate - round(p * 128) / 128) < 3e-3 and same > 0.9995 and bool((ey <= tol).all())
        if p == 0:
            d8, dv = outs['8w'][0].float(), outs['ov'][0].float()
            print('   ov vs 8w: identical', float((d8 == dv).float().mean()), 'max |diff|', float((d8 - dv).abs().max()))
        if args.time:
            outs['ov-noepi'] = outs['ov']
            for name, kern, diag in (('8w', 2, 128), ('ov', 4, 0), ('ov-noepi', 4, 1)):
                d = outs[name][2]
                for _ in range(3):
                    tg(ctypes.byref(d), st, kern, 0, diag)
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        tg(ctypes.byref(d), st, kern, 0, diag)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / args.iters * 1e3)
                ts.sort()
                print(f'   time p={p} {name}: median {ts[2]:.1f} us  min {ts[0]:.1f} us  ({2.0 * M * N * K / ts[2] / 1e6 / 2500 * 100:.1f} % of 2.5 PF)')
    print('OV_CHECK', 'PASS' if ok else 'FAIL')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
