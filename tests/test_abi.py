"""CPU: the C-ABI library loads and exports every symbol include/ecgvit_hip.h declares; the ctypes table covers them."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
import ecg_representation_learning_amd as E


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'ecgvit_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ecgvit_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(E.hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(E.hip.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/ecgvit_hip.h but not exported'
    assert set(syms) == set(E.hip.SIGNATURES), set(syms) ^ set(E.hip.SIGNATURES)


def test_tools_library_exports_what_its_header_declares_and_product_does_not():
    """tools/ecgvit_hip_tools.h (probes, the one-item attention backward, stamps, A/B switches) lives in build/libecgvit_hip_tools.so only: the
    product library exports none of it, the product binding names none of it"""
    src = open(os.path.join(ROOT, 'tools', 'ecgvit_hip_tools.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    tsyms = sorted(set(re.findall(r'\b(ecgvit_[a-z0-9_]+)\s*\(', src)))
    assert 'ecgvit_probe_mfma_layout' in tsyms and 'ecgvit_attention_bwd_oneitem' in tsyms and 'ecgvit_debug_attn_stamps' in tsyms
    tools_path = os.path.join(os.path.dirname(E.hip.LIB_PATH), 'csrc', 'build', 'libecgvit_hip_tools.so')
    if not os.path.exists(tools_path):
        import __graft_entry__
        __graft_entry__.build()
    tl, pl = ctypes.CDLL(tools_path), ctypes.CDLL(E.hip.LIB_PATH)
    for s_ in tsyms:
        assert hasattr(tl, s_), f'{s_} declared in tools/ecgvit_hip_tools.h but not exported by the tools library'
        assert not hasattr(pl, s_), f'{s_} is a tools entry point but the PRODUCT library exports it'
        assert s_ not in E.hip.SIGNATURES
    for s_ in declared_symbols():          # the tools library is the product ABI plus the above
        assert hasattr(tl, s_)


def test_version_and_abi():
    l = E.hip.lib()
    assert l.ecgvit_abi_version() == 6
    assert b'gfx950' in l.ecgvit_version()
    assert l.ecgvit_version().decode().endswith(f'abi{l.ecgvit_abi_version()}')   # the two identity calls agree


def test_gemm_desc_matches_header_field_order():
    src = open(os.path.join(ROOT, 'include', 'ecgvit_hip.h')).read()
    body = src[src.index('typedef struct ecgvit_gemm_desc {'):src.index('} ecgvit_gemm_desc;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = []
    for stmt in body.split('{', 1)[1].split(';'):
        stmt = stmt.strip()
        if not stmt:
            continue
        decl = stmt.split(',')
        for i, d in enumerate(decl):
            names.append(re.findall(r'([A-Za-z_0-9]+)\s*$', d.strip())[0])
    assert names == [f[0] for f in E.hip.GemmDesc._fields_]


def test_gemm_kernel_route_query_is_host_only():
    """ecgvit_gemm_kernel answers which kernel family ecgvit_gemm would launch -- pure host logic (nothing is launched, no pointer is
    dereferenced), so it runs without a GPU: the dispatch bench.py's probe relies on, pinned per launch type of the train step"""
    hip = E.hip
    l = hip.lib()

    def route(layout, dtype, out_dtype, M, N, K, lda, ldb, ldc, epilogue=0, aux=False, res=False, ws=False):
        d = hip.GemmDesc()
        d.layout, d.dtype, d.out_dtype, d.epilogue = layout, dtype, out_dtype, epilogue
        d.M, d.N, d.K, d.batch1, d.batch2 = M, N, K, 1, 1
        d.A, d.B, d.C = 0x10000000, 0x20000000, 0x30000000           # never dereferenced: alignment is all that is looked at
        d.lda, d.ldb, d.ldc = lda, ldb, ldc
        d.bias = 0x40000000
        d.alpha = 1.0
        if aux:
            d.aux, d.ldaux = 0x50000000, N
        if res:
            d.residual, d.ldr = 0x60000000, N
        if ws:
            d.workspace, d.workspace_bytes, d.colsum_out = 0x70000000, 1 << 30, 0x78000000
        return l.ecgvit_gemm_kernel(ctypes.byref(d))
    M, dm, f = 512 * 251, 768, 3072
    UP = hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_GELU_GRAD_AUX | hip.EPI_DROPOUT
    assert route(hip.GEMM_NT, hip.BF16, hip.BF16, M, 3 * dm, dm, dm, dm, 3 * dm) == hip.KERNEL_GEMM_NT                       # QKV forward
    assert route(hip.GEMM_NT, hip.BF16, hip.BF16, M, f, dm, dm, dm, f, epilogue=UP, aux=True) == hip.KERNEL_GEMM_NT            # FFN-up forward
    assert route(hip.GEMM_NT, hip.BF16, hip.BF16, M, f, dm, dm, dm, f, epilogue=hip.EPI_MUL_AUX | hip.EPI_COLSUM, aux=True, ws=True) == hip.KERNEL_GEMM_NT
    assert route(hip.GEMM_NT, hip.BF16, hip.BF16, 2000, dm, dm, dm, dm, dm) == hip.KERNEL_GEMM_BF16                             # short batch: 128^2 kernel
    assert route(hip.GEMM_NT, hip.BF16, hip.BF16, M // 251 * 250, dm, 240, 240, 240, dm, epilogue=hip.EPI_BIAS) == hip.KERNEL_GEMM_BF16   # patch embed (K = 240)
    assert route(hip.GEMM_TN, hip.BF16, hip.F32, 3 * dm, dm, M, 3 * dm, dm, dm, ws=True) == hip.KERNEL_GEMM_WGRAD                 # QKV weight gradient
    assert route(hip.GEMM_TN, hip.BF8_E5M2, hip.F32, 3072, 1024, 256 * 501, 3072, 1024, 1024, ws=True) == hip.KERNEL_GEMM_WGRAD   # 8-bit weight gradient
    assert route(hip.GEMM_NT, hip.FP8_E4M3, hip.BF16, 256 * 501, 4096, 1024, 1024, 1024, 4096) == hip.KERNEL_GEMM_NT              # 8-bit forward
    assert route(hip.GEMM_NT, hip.FP8_E4M3, hip.BF16, 1000, 4096, 1024, 1024, 1024, 4096) == hip.KERNEL_NONE                      # 8-bit: large shapes only
    assert route(hip.GEMM_NN, hip.F32, hip.F32, 100, 64, 32, 32, 64, 64) == hip.KERNEL_GEMM_F32
    assert route(hip.GEMM_NT, hip.BF16, hip.BF16, M, dm + 4, dm, dm, dm, dm + 4) == hip.KERNEL_NONE                              # N % 8 != 0: rejected
    assert l.ecgvit_gemm_kernel(None) == hip.KERNEL_NONE


def test_binding_constants_are_the_headers():
    """every numeric #define of include/ecgvit_hip.h that the Python binding restates (dtype / layout / epilogue / kernel-family codes, error codes)
    has the same value there: `ECGVIT_X` <-> `hip.X`"""
    hip = E.hip
    text = open(os.path.join(ROOT, 'include', 'ecgvit_hip.h')).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r'^#define ECGVIT_([A-Z0-9_]+)\s+(-?\d+)\b', text, re.M)}
    assert len(defs) >= 25, defs
    checked = 0
    for name, val in defs.items():
        if hasattr(hip, name):
            assert getattr(hip, name) == val, (name, val, getattr(hip, name))
            checked += 1
    for need in ('F32', 'BF16', 'FP8_E4M3', 'BF8_E5M2', 'GEMM_NT', 'GEMM_NN', 'GEMM_TN', 'EPI_BIAS', 'EPI_GELU', 'EPI_GELU_BWD', 'EPI_RESIDUAL', 'EPI_ACCUM',
                 'EPI_DROPOUT', 'EPI_COLSUM', 'EPI_GELU_GRAD_AUX', 'EPI_MUL_AUX', 'EPI_QUANT_OUT', 'EPI_NO_OUT', 'KERNEL_NONE', 'KERNEL_GEMM_F32',
                 'KERNEL_GEMM_BF16', 'KERNEL_GEMM_NT', 'KERNEL_GEMM_WGRAD'):
        assert need in defs and hasattr(hip, need), need
    assert checked >= 23
    assert {1: 'EINVAL', 2: 'ELAUNCH'} == {defs['EINVAL']: 'EINVAL', defs['ELAUNCH']: 'ELAUNCH'} and set(hip._ERR) == {1, 2}
