#!/usr/bin/env python3
"""A/B of the output-store policy of the PLAIN 8-bit A.B^T products at the EcgVit-large shapes (M = 256 x 501 token rows): default-policy against
non-temporal stores (ecgvit_tools_gemm diag 256 / 512), one process, interleaved.  usage: python tools/fp8_nt_ab.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecg_representation_learning_amd import hip  # noqa: E402
hip.use_library(os.path.join(ROOT, 'ecg-representation-learning_amd', 'csrc', 'build', 'libecgvit_hip_tools.so'))
import bench as _bench  # noqa: E402
print('kernel_source_sha16:', _bench.kernel_source_hash(), flush=True)


def main():
    lib = hip.lib()
    tg = lib.ecgvit_tools_gemm
    tg.restype, tg.argtypes = ctypes.c_int, [ctypes.POINTER(hip.GemmDesc), ctypes.c_void_p] + [ctypes.c_int] * 3
    M, d, f = 256 * 501, 1024, 4096
    one = torch.ones(1, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    for name, K, N, fmt, tdt in (('fwd qkv', d, 3 * d, hip.FP8_E4M3, torch.float8_e4m3fn), ('dgrad qkv', 3 * d, d, hip.BF8_E5M2, torch.float8_e5m2),
                                 ('dgrad out', d, d, hip.BF8_E5M2, torch.float8_e5m2), ('dgrad ffn_up', f, d, hip.BF8_E5M2, torch.float8_e5m2)):
        X8 = torch.randn(M, K, device='cuda').to(tdt).view(torch.uint8)
        W8 = (torch.randn(N, K, device='cuda') * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
        C = {v: torch.empty(M, N, device='cuda', dtype=torch.bfloat16) for v in (256, 512)}
        descs = {v: hip.gemm_desc(hip.GEMM_NT, X8, W8, C[v], M, N, K, K, K, N, fp8_format=fmt, scale_a=one, scale_b=one) for v in (256, 512)}
        t = {256: [], 512: []}
        for v in (256, 512):
            assert tg(ctypes.byref(descs[v]), st, 2, 0, v) == 0
        torch.cuda.synchronize()
        assert torch.equal(C[256], C[512])
        for _ in range(6):
            for v in (256, 512):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    tg(ctypes.byref(descs[v]), st, 2, 0, v)
                e1.record()
                torch.cuda.synchronize()
                t[v].append(e0.elapsed_time(e1) / 8 * 1e3)
        fl = 2.0 * M * N * K
        a, b = sorted(t[256][1:])[2], sorted(t[512][1:])[2]
        print(f'{name:13s} K={K:4d} N={N:4d} out {M * N * 2 / 2 ** 20:5.0f} MB: default {a:7.1f} us ({fl / a / 1e6:5.0f} TF)   non-temporal {b:7.1f} us ({fl / b / 1e6:5.0f} TF)  {100 * (b / a - 1):+.1f} %', flush=True)


if __name__ == '__main__':
    main()
