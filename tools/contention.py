#!/usr/bin/env python3
"""What a large A.B^T launch costs while another kernel holds CUs (the situation of the data-parallel backward pass: RCCL all-reduce
kernels overlapped with the GEMMs).  A stand-in kernel occupies n CUs on a side stream for longer than the GEMMs run; the GEMMs are
timed with tiles_per_workgroup = 0 (persistent, static shares) and k > 0 (dispatcher-balanced chunks).  Tools build."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ecg_representation_learning_amd as E  # noqa: E402,F401
from ecg_representation_learning_amd import hip  # noqa: E402
hip.use_library(os.path.join(ROOT, 'ecg-representation-learning_amd', 'csrc', 'build', 'libecgvit_hip_tools.so'))   # explicit: the diagnostic build, never the shipped library


def main():
    lib = hip.lib()
    occ = lib.ecgvit_tools_occupy
    occ.restype, occ.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    M, bf = 512 * 251, torch.bfloat16
    side = torch.cuda.Stream()
    done = torch.zeros(1, dtype=torch.int32, device='cuda')
    for name, K, N in (('qkv', 768, 2304), ('out', 768, 768), ('ffn_down', 3072, 768)):
        X = torch.randn(M, K, device='cuda').to(bf)
        W = (torch.randn(N, K, device='cuda') * 0.03).to(bf)
        C = torch.empty(M, N, device='cuda', dtype=bf)
        ref = None
        for held in (0, 8, 16, 32):
            row = []
            for tpw in (0, 1, 2, 4):
                for _ in range(2):
                    hip.gemm(hip.GEMM_NT, X, W, C, M, N, K, K, K, N, tiles_per_workgroup=tpw)
                torch.cuda.synchronize()
                done.zero_()
                if held:
                    occ(held, int(40e6), done.data_ptr(), side.cuda_stream)      # ~20 ms of shader cycles
                    torch.cuda._sleep(200000)                                   # let the holders get their CUs first
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    hip.gemm(hip.GEMM_NT, X, W, C, M, N, K, K, K, N, tiles_per_workgroup=tpw)
                e1.record()
                torch.cuda.synchronize()
                if ref is None:
                    ref = C.clone()
                assert torch.equal(C, ref), (name, held, tpw)
                assert int(done) == held
                row.append(f'tpw={tpw}: {e0.elapsed_time(e1) / 5 * 1e3:7.1f} us')
            print(f'{name:9s} K={K} N={N}  {held:2d} CUs held | ' + ' | '.join(row), flush=True)


if __name__ == '__main__':
    main()
