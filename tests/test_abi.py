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


def test_version_and_abi():
    l = E.hip.lib()
    assert l.ecgvit_abi_version() == 3
    assert b'gfx950' in l.ecgvit_version()


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
