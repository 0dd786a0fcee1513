#!/usr/bin/env python3
"""the two FFN-wide emitting launches of the fp8 path at the EcgVit-large shapes (M = 256 x 501, d = 1024, f = 4096), writing and no-output forms, next to
the plain 8-bit products of the same shapes: python tools/fp8_epilogue_bench.py [lib.so]   (lib.so: another build of the library to time instead)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecg_representation_learning_amd import hip  # noqa: E402
if len(sys.argv) > 1:
    hip.use_library(sys.argv[1])
import bench as _bench  # noqa: E402
print('kernel_source_sha16:', _bench.kernel_source_hash(), '(the shipped sources; a library named on the command line may be another build)', flush=True)
M, d, f = 256 * 501, 1024, 4096
bf = torch.bfloat16
g = torch.Generator().manual_seed(1)
one = torch.tensor([0.5], device='cuda'); qs = torch.tensor([0.02], device='cuda')
X8 = torch.randn(M, d, generator=g).to(torch.float8_e4m3fn).cuda().view(torch.uint8)
W18 = (torch.randn(f, d, generator=g) * 0.3).to(torch.float8_e4m3fn).cuda().view(torch.uint8)       # FFN-up weight [f, d]
G8 = torch.randn(M, d, generator=g).to(torch.float8_e5m2).cuda().view(torch.uint8)                  # dY of the FFN-down site
W2T8 = (torch.randn(f, d, generator=g) * 0.3).to(torch.float8_e4m3fn).cuda().view(torch.uint8)      # transposed FFN-down weight [f, d]
bias = torch.randn(f, device='cuda')
C = torch.empty(M, f, device='cuda', dtype=bf); aux = torch.empty(M, f, device='cuda', dtype=bf)
auxin = (torch.rand(M, f, device='cuda') * 1.2).to(bf)
q8 = torch.empty(M, f, dtype=torch.uint8, device='cuda'); am = torch.zeros(1, device='cuda'); cs = torch.zeros(f, device='cuda')
ws = torch.empty(max(hip.lib().ecgvit_colsum_workspace(M, f), 8 * ((M + 255) // 256) * f), dtype=torch.uint8, device='cuda')
UP = hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_GELU_GRAD_AUX | hip.EPI_DROPOUT
DH = hip.EPI_MUL_AUX | hip.EPI_COLSUM
Q = dict(q8_out=q8, ldq8=f, q8_scale=qs, q8_amax=am)
cases = {
    'FFN-up plain 8-bit': lambda: hip.gemm(hip.GEMM_NT, X8, W18, C, M, f, d, d, d, f, fp8_format=hip.FP8_E4M3, scale_a=one, scale_b=one),
    'FFN-up GELU + mask (first pass)': lambda: hip.gemm(hip.GEMM_NT, X8, W18, C, M, f, d, d, d, f, fp8_format=hip.FP8_E4M3, scale_a=one, scale_b=one, epilogue=UP, bias=bias, aux=aux, ldaux=f,
                                                        dropout_p=0.1, seed=3),
    'FFN-up emitting, writing form': lambda: hip.gemm(hip.GEMM_NT, X8, W18, C, M, f, d, d, d, f, fp8_format=hip.FP8_E4M3, scale_a=one, scale_b=one, epilogue=UP | hip.EPI_QUANT_OUT, bias=bias,
                                                      aux=aux, ldaux=f, dropout_p=0.1, seed=3, q8_format=hip.FP8_E4M3, **Q),
    'FFN-up emitting, no-output form': lambda: hip.gemm(hip.GEMM_NT, X8, W18, None, M, f, d, d, d, f, fp8_format=hip.FP8_E4M3, scale_a=one, scale_b=one,
                                                        epilogue=UP | hip.EPI_QUANT_OUT | hip.EPI_NO_OUT, bias=bias, aux=aux, ldaux=f, dropout_p=0.1, seed=3, q8_format=hip.FP8_E4M3, **Q),
    'FFN-down dgrad plain 8-bit': lambda: hip.gemm(hip.GEMM_NT, G8, W2T8, C, M, f, d, d, d, f, fp8_format=hip.BF8_E5M2, scale_a=one, scale_b=one),
    'FFN-down dgrad x aux + colsum (first pass)': lambda: hip.gemm(hip.GEMM_NT, G8, W2T8, C, M, f, d, d, d, f, fp8_format=hip.BF8_E5M2, scale_a=one, scale_b=one, epilogue=DH, aux=auxin, ldaux=f,
                                                                   workspace=ws, colsum_out=cs),
    'FFN-down dgrad emitting, writing form': lambda: hip.gemm(hip.GEMM_NT, G8, W2T8, C, M, f, d, d, d, f, fp8_format=hip.BF8_E5M2, scale_a=one, scale_b=one, epilogue=DH | hip.EPI_QUANT_OUT,
                                                              aux=auxin, ldaux=f, workspace=ws, colsum_out=cs, q8_format=hip.BF8_E5M2, **Q),
    'FFN-down dgrad emitting, no-output form': lambda: hip.gemm(hip.GEMM_NT, G8, W2T8, None, M, f, d, d, d, f, fp8_format=hip.BF8_E5M2, scale_a=one, scale_b=one,
                                                                epilogue=DH | hip.EPI_QUANT_OUT | hip.EPI_NO_OUT, aux=auxin, ldaux=f, workspace=ws, colsum_out=cs, q8_format=hip.BF8_E5M2, **Q),
}
times = {k: [] for k in cases}
for fn in cases.values():
    fn(); fn()
torch.cuda.synchronize()
for _ in range(5):
    for k, fn in cases.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) * 200)
fl = 2.0 * M * d * f
for k in cases:
    t = sorted(times[k])
    print(f'{k:46s}: median {t[len(t) // 2]:8.1f} us  min {t[0]:8.1f} us  {fl / t[len(t) // 2] / 1e6:7.0f} TFLOP/s', flush=True)
