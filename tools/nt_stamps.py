#!/usr/bin/env python3
"""In-kernel cycle anatomy of gemm_nt_kernel (tools build, stamped instantiations): per launch shape the clock the chip holds
(delta s_memtime / delta s_memrealtime x 100 MHz), main-loop cycles per 64-deep K-tile and epilogue cycles per output tile
(median over the 256 workgroups).  Shares only -- the stamped build is not the timed build."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ecg_representation_learning_amd as E  # noqa: E402,F401
from ecg_representation_learning_amd import hip  # noqa: E402
hip.use_library(os.path.join(ROOT, 'ecg-representation-learning_amd', 'csrc', 'build', 'libecgvit_hip_tools.so'))   # explicit: the diagnostic build, never the shipped library
from ecg_representation_learning_amd.hip import (EPI_BIAS, EPI_GELU, EPI_DROPOUT, EPI_COLSUM, EPI_GELU_GRAD_AUX, EPI_MUL_AUX, GEMM_NT)  # noqa: E402


COLD = '--cold' in sys.argv
KERNEL = int(os.environ.get('NT_STAMP_KERNEL', '2'))   # 3 = the four-wave body (plain products only; diag 1 = stamped, +2 = nt stores, +4 = stores dropped, +8 = no epilogue)
variants = [(0, int(x)) for x in os.environ.get('NT_STAMP_DIAGS', '1,3').split(',')]   # (raster_g, diag): diag 1 = stamped build, +2 = output stores dropped, +4 = no DMA after the prologue, +8 = no counted waits


def main():
    global ts
    lib = hip.lib()
    tg = lib.ecgvit_tools_gemm
    tg.restype, tg.argtypes = ctypes.c_int, [ctypes.POINTER(hip.GemmDesc), ctypes.c_void_p] + [ctypes.c_int] * 3
    ts = lib.ecgvit_tools_nt_stamps
    ts.restype, ts.argtypes = ctypes.c_int, [ctypes.c_void_p]
    M, d, f = 512 * 251, 768, 3072
    UP = EPI_BIAS | EPI_GELU | EPI_GELU_GRAD_AUX | EPI_DROPOUT
    DH = EPI_MUL_AUX | EPI_COLSUM
    cases = [('qkv plain [L2]', d, 3 * d, 0), ('ffn_down plain [L2]', f, d, 0), ('qkv plain', d, 3 * d, 0), ('out plain', d, d, 0), ('ffn_up plain', d, f, 0), ('ffn_down plain', f, d, 0), ('ffn_up GELU', d, f, UP), ('ffn_down dgrad', d, f, DH)]
    bf, dev = torch.bfloat16, 'cuda'
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for name, K, N, epi in cases:
        if KERNEL == 3 and epi:
            continue
        X = torch.randn(M, K, device=dev).to(bf)
        W = (torch.randn(N, K, device=dev) * 0.03).to(bf)
        C = torch.empty(M, N, device=dev, dtype=bf)
        bias = torch.randn(N, device=dev)
        aux = (torch.rand(M, N, device=dev) * 1.2).to(bf) if epi else None
        cso = torch.zeros(N, device=dev) if epi & EPI_COLSUM else None
        lda = ldb = K
        if name.endswith('[L2]'):       # overlapping rows (row pitch 128 B): 12-24 rows share every line -> operands always hit L2
            lda = ldb = 64
            X = torch.randn(M * 64 + K, device=dev).to(bf)
            W = (torch.randn(N * 64 + K, device=dev) * 0.03).to(bf)
        desc = hip.gemm_desc(GEMM_NT, X, W, C, M, N, K, lda, ldb, N, epilogue=epi, bias=bias if epi & EPI_BIAS else None, aux=aux, ldaux=N,
                             dropout_p=0.1 if epi & EPI_DROPOUT else 0.0, seed=7, workspace=ws, colsum_out=cso)
        st = torch.cuda.current_stream().cuda_stream
        reps = max(20, int(1.0 / 0.0006))
        if COLD:
            # every launch from cold caches, as inside the train step: a 1-GiB fill between launches evicts L2 and the Infinity Cache
            flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            for (g, diag) in variants:
                rows, walls = [], []
                for _ in range(12):
                    flush.fill_(1)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    assert tg(ctypes.byref(desc), st, KERNEL, g, diag) == 0
                    e1.record()
                    torch.cuda.synchronize()
                    walls.append(e0.elapsed_time(e1) * 1e3)
                    rows.append(stamps())
                report(name + f' COLD g={g} diag={diag} wall {np.median(walls):6.1f} us', K, N, np.median(np.stack(rows), axis=0))
            continue
        for (g, diag) in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                rc = tg(ctypes.byref(desc), st, KERNEL, g, diag)
                assert rc == 0, rc
            e1.record()
            torch.cuda.synchronize()
            report(name + f' g={g} diag={diag} wall {e0.elapsed_time(e1) / reps * 1e3:6.1f} us', K, N)


def stamps():
    buf = np.zeros(256 * 8, dtype=np.uint64)
    assert ts(buf.ctypes.data) == 0
    s = buf.reshape(256, 8).astype(np.float64)
    s[:, 2] -= s[:, 0]; s[:, 3] -= s[:, 1]; s[:, 0] = 0; s[:, 1] = 0   # durations, so that launches can be averaged
    return s


def report(name, K, N, s=None):
    if True:
        if s is None:
            s = stamps()
        clk = (s[:, 2] - s[:, 0]) / (s[:, 3] - s[:, 1]) * 0.1
        per_k = s[:, 4] / (s[:, 6] * s[:, 7])
        per_e = s[:, 5] / s[:, 6]
        tot = s[:, 2] - s[:, 0]
        print(f'{name:52s} K={K:4d} N={N:4d}: clock {np.median(clk):.3f} GHz (min {clk.min():.3f} max {clk.max():.3f}); main loop {np.median(per_k):7.0f} cyc/K-tile '
              f'(MFMA alone 2048); epilogue {np.median(per_e):7.0f} cyc/tile; tiles/WG {np.median(s[:, 6]):.0f}; kernel {np.median(tot):9.0f} cyc '
              f'= main {100 * np.median(s[:, 4] / tot):4.1f} % + epi {100 * np.median(s[:, 5] / tot):4.1f} %', flush=True)


if __name__ == '__main__':
    main()
