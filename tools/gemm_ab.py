#!/usr/bin/env python3
"""A/B timing of the large A.B^T kernels on the eight Linear products of one EcgVit-base layer (M = 512*251 token rows), WITH the
epilogues the train step uses, all variants interleaved in ONE process on ONE device (guide rule 24).

Needs the tools build (`make -C ecg-representation-learning_amd/csrc tools`): `ecgvit_tools_gemm(desc, stream, kernel, raster_g)`
kernel 1 = retired LDS-patch kernel (gemm_bf16_q_kernel), 2 = gemm_nt_kernel's dispatch (diag 128: eight-wave body only, 256 / 512: default-policy /
non-temporal output stores), 3 = gemm_nt_kernel_4w (diag 2: non-temporal stores); `lib` = torch.matmul (hipBLASLt), plain product only.
usage: python tools/gemm_ab.py [--rounds 5] [--iters 10] [--groups 0,1,3] [--check]
"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ecg_representation_learning_amd as E  # noqa: E402,F401
from ecg_representation_learning_amd import hip  # noqa: E402
hip.use_library(os.path.join(ROOT, 'ecg-representation-learning_amd', 'csrc', 'build', 'libecgvit_hip_tools.so'))   # explicit: the diagnostic build, never the shipped library
import bench as _bench  # noqa: E402
print('kernel_source_sha16:', _bench.kernel_source_hash(), '(sources of the library build measured: tools/check_profiles.py holds committed tables to the round\'s bench line)', flush=True)
from ecg_representation_learning_amd.hip import (EPI_BIAS, EPI_GELU, EPI_RESIDUAL, EPI_DROPOUT, EPI_COLSUM,  # noqa: E402
                                                  EPI_GELU_GRAD_AUX, EPI_MUL_AUX, GEMM_NT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--m', type=int, default=512 * 251)
    ap.add_argument('--dim', type=int, default=768, help='hidden size (FFN width = 4 x): 512 with --m 64256 = EcgVit-small at 256 records')
    ap.add_argument('--groups', default='0', help='comma list of raster_g values for kernel 2 (0 = built-in choice)')
    ap.add_argument('--check', action='store_true', help='compare kernel 2 with kernel 1 and with an f32 reference on a row sample')
    ap.add_argument('--no-old', action='store_true')
    ap.add_argument('--no-lib', action='store_true')
    ap.add_argument('--only', default='')
    ap.add_argument('--plain', action='store_true', help='also time every epilogue case with epilogue 0')
    ap.add_argument('--json', default='')
    ap.add_argument('--aux-ld0', action='store_true', help='diagnostics: aux row pitch 0 (every row reads / writes ONE cache-resident row): what the aux stream costs')
    ap.add_argument('--aux8', action='store_true', help='the FFN-wide epilogues with the e4m3 saved tensor (ECGVIT_EPI_AUX8)')
    ap.add_argument('--rdv', action='store_true', help='experiment: the stamped eight-wave instantiation without / with the XCD rendezvous per tile round (plain, FFN-up, x-aux cases)')
    ap.add_argument('--nt4', action='store_true', help='also time the four-wave body (kernel 3) on every plain product')
    ap.add_argument('--rowaffine', action='store_true', help='LayerNorm-fold pricing (round 6): the QKV forward and (with --aux8) the FFN-up forward with the row-affine epilogue '
                    'v a[m] + (b[m] g[n] + c[n]) of LN folded into its consumer product, next to what ships, and the LayerNorm forward pass each would replace')
    args = ap.parse_args()
    lib = hip.lib()
    tg = lib.ecgvit_tools_gemm
    tg.restype, tg.argtypes = ctypes.c_int, [ctypes.POINTER(hip.GemmDesc), ctypes.c_void_p] + [ctypes.c_int] * 3
    M, d, f = args.m, args.dim, 4 * args.dim
    LIN = EPI_BIAS | EPI_RESIDUAL | EPI_DROPOUT
    A8 = hip.EPI_AUX8 if args.aux8 else 0
    UP = EPI_BIAS | EPI_GELU | EPI_GELU_GRAD_AUX | EPI_DROPOUT | A8
    DH = EPI_MUL_AUX | EPI_COLSUM | A8
    cases = [('fwd qkv', d, 3 * d, 0), ('fwd out', d, d, LIN), ('fwd ffn_up', d, f, UP), ('fwd ffn_down', f, d, LIN),
             ('dgrad qkv', 3 * d, d, 0), ('dgrad out', d, d, 0), ('dgrad ffn_up', f, d, 0), ('dgrad ffn_down', d, f, DH)]
    if args.plain:
        cases += [(n + ' [plain]', K, N, 0) for (n, K, N, e_) in cases if e_]
    if args.only:
        cases = [c for c in cases if args.only in c[0]]
    groups = [int(g) for g in args.groups.split(',')]
    bf = torch.bfloat16
    dev = 'cuda'
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    results = []
    if args.rowaffine:
        ra = lib.ecgvit_tools_rowaffine
        ra.restype, ra.argtypes = ctypes.c_int, [ctypes.c_void_p] * 3
        torch.manual_seed(2)
        row_a, row_b = torch.rand(M, device=dev) + 0.5, torch.randn(M, device=dev) * 0.1
        col_g = torch.randn(max(3 * d, f), device=dev)
        assert ra(row_a.data_ptr(), row_b.data_ptr(), col_g.data_ptr()) == 0
        # the pass the fold would delete: LayerNorm forward over [M, d] bf16 (writes xn, mean, rstd)
        x = torch.randn(M, d, device=dev).to(bf)
        y = torch.empty_like(x)
        gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        st0 = torch.cuda.current_stream().cuda_stream
        ts = []
        for _ in range(args.rounds + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                lib.ecgvit_layernorm_fwd(x.data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, d, 1e-5, hip.BF16, st0)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / args.iters * 1e3)
        ts = sorted(ts[1:])
        print(f'layernorm_fwd [{M} x {d}] bf16 (the pass a fold deletes, once per consumer): median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us', flush=True)
    for name, K, N, epi in cases:
        torch.manual_seed(1)
        X = torch.randn(M, K, device=dev).to(bf)
        W = (torch.randn(N, K, device=dev) * 0.03).to(bf)
        bias = torch.randn(N, device=dev) * 0.1
        res = torch.randn(M, N, device=dev).to(bf) if epi & EPI_RESIDUAL else None
        aux = (torch.rand(M, N, device=dev) * 1.2).to(bf) if epi & (EPI_GELU | EPI_MUL_AUX) else None
        if aux is not None and args.aux8:
            aux = aux.to(torch.float8_e4m3fn).view(torch.uint8)
        cso = torch.zeros(N, device=dev) if epi & EPI_COLSUM else None
        outs = {}

        def make(C, auxbuf):
            return hip.gemm_desc(GEMM_NT, X, W, C, M, N, K, K, K, N, epilogue=epi, bias=bias if epi & EPI_BIAS else None, residual=res,
                                 ldr=N, aux=auxbuf, ldaux=0 if args.aux_ld0 else N, dropout_p=0.1 if epi & EPI_DROPOUT else 0.0, seed=1234, workspace=ws,
                                 colsum_out=cso)

        variants = []
        if not args.no_old:
            variants.append(('old', 1, 0, 0))
        for g in groups:
            variants.append((f'8w g={g}', 2, g, 128 | 256))          # eight-wave body, default-policy stores
            if epi == 0:
                variants.append((f'8w g={g} nt', 2, g, 128 | 512))   # eight-wave body, non-temporal stores
            variants.append((f'shipped g={g}', 2, g, 0))             # what the library's dispatch picks
        if args.rdv and epi in (0, UP, DH) and not args.aux8:
            variants.append(('8w stamped', 2, 0, 1))
            variants.append(('8w stamped rdv', 2, 0, 33))
        if args.nt4 and epi == 0:
            variants.append(('4w', 3, 0, 0))
            variants.append(('4w nt', 3, 0, 2))
        if args.nt4 and epi in (LIN, DH):
            variants.append(('4w', 3, 0, 0))
        fold = None
        if args.rowaffine and (name == 'fwd qkv' or (name == 'fwd ffn_up' and args.aux8)):
            fold = ('8w nt + LN fold' if epi == 0 else 'shipped + LN fold', 2, 0, 1024)
            variants.append(fold)
        C = {v[0]: torch.empty(M, N, device=dev, dtype=bf) for v in variants}
        A = {v[0]: (aux.clone() if aux is not None else None) for v in variants}
        descs = {v[0]: make(C[v[0]], A[v[0]]) for v in variants}
        if fold is not None and epi == 0:   # the QKV forward's fold body takes its second column vector through the bias
            descs[fold[0]] = hip.gemm_desc(GEMM_NT, X, W, C[fold[0]], M, N, K, K, K, N, epilogue=EPI_BIAS, bias=bias, workspace=ws)
        st = torch.cuda.current_stream().cuda_stream

        def run(v):
            rc = tg(ctypes.byref(descs[v[0]]), st, v[1], v[2], v[3])
            if rc:
                raise RuntimeError(f'{name} {v[0]}: rc={rc}')

        fns = [(v[0], (lambda v=v: run(v))) for v in variants]
        if not args.no_lib:
            Cl = torch.empty(M, N, device=dev, dtype=bf)
            fns.append(('lib plain', lambda: torch.matmul(X, W.t(), out=Cl)))
        times = {n: [] for n, _ in fns}
        for n, fn in fns:
            for _ in range(2):
                fn()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for n, fn in fns:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) / args.iters * 1e3)
        fl = 2.0 * M * K * N
        for n, _ in fns:
            t = sorted(times[n])
            med, mn = t[len(t) // 2], t[0]
            print(f'{name:15s} K={K:4d} N={N:4d} epi={epi:3d}  {n:16s}: median {med:7.1f} us  min {mn:7.1f} us  {fl / med / 1e6:7.1f} TFLOP/s '
                  f'({100 * fl / med / 1e6 / 2500:4.1f} %)', flush=True)
            results.append(dict(case=name, K=K, N=N, epilogue=epi, variant=n, median_us=med, min_us=mn, tflops=fl / med / 1e6))
        if fold is not None and args.check:
            # the fold body against the same transform applied to the shipped body's output path: v a + (b g + c) on the f32 product of a row sample
            rows = torch.randint(0, M, (256,), device=dev)
            acc = X[rows].float() @ W.float().t()
            if epi == 0:
                want = acc * row_a[rows, None] + (row_b[rows, None] * col_g[None, :N] + bias[None, :])
                err = (C[fold[0]][rows].float() - want).abs().max().item()
                print(f'   check {fold[0]} vs f32 reference (256 rows): max abs err {err:.3e} (|want| max {want.abs().max().item():.2f})', flush=True)
        if args.check and len(variants) >= 2:
            ref = C[variants[0][0]].float()
            for v in variants[1:]:
                dlt = (C[v[0]].float() - ref).abs()
                same = (C[v[0]] == C[variants[0][0]]).float().mean().item()
                msg = f'   check {v[0]}: max|new-old| {dlt.max().item():.3e}  identical {100 * same:.4f} %'
                if aux is not None and epi & EPI_GELU:
                    msg += f'  aux identical {100 * (A[v[0]] == A[variants[0][0]]).float().mean().item():.4f} %'
                print(msg, flush=True)
        if args.check:
            # f32 reference on a row sample (plain product + bias only cases get the full check elsewhere: tests/test_gpu_ops.py)
            rows = torch.randint(0, M, (512,), device=dev)
            acc = X[rows].float() @ W.float().t()
            if epi == 0:
                for v in variants:
                    err = (C[v[0]][rows].float() - acc).abs().max().item()
                    print(f'   check {v[0]} vs f32 reference (512 rows): max abs err {err:.3e}', flush=True)
    if args.json:
        with open(args.json, 'w') as fh:
            json.dump(results, fh, indent=1)


if __name__ == '__main__':
    main()
