"""CPU: resources of the shipped library's kernels, read from the code-object metadata (tools/code_objects.py; llvm-readelf, no GPU).
The streaming kernels retire their LDS-DMA pieces with COUNTED `s_waitcnt vmcnt(N)`: a spilled register would add scratch loads / stores to
that count (and to the CU's memory path those kernels are bound by), so every instantiation the train step can launch must be spill-free."""
import os
import re
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'tools'))
LIB = os.path.join(ROOT, 'ecg-representation-learning_amd', 'libecgvit_hip.so')


@pytest.fixture(scope='module')
def kernels():
    import code_objects
    if not os.path.exists(code_objects.READELF):
        pytest.skip('llvm-readelf not in this image')
    ks = code_objects.kernels(LIB)
    assert len(ks) > 100, len(ks)
    return ks


def test_streaming_kernels_do_not_spill(kernels):
    counted = ('attn_bwd_pers_kernel', 'attn_fwd_bf16_kernel', 'attn_fwd_stream_kernel', 'gemm_nt_kernel_4w', 'gemm_wgrad_kernel', 'gemm_wgrad8_kernel', 'gemm_nt_kernel')
    seen = 0
    for name, k in kernels.items():
        if not any(c in name for c in counted):
            continue
        if re.search(r'gemm_nt_kernelI\w+?Lin1E', name):     # FLAGS = -1: the run-time-flag fallback bodies (no train-step launch takes them)
            continue
        seen += 1
        assert k['vgpr_spill_count'] == 0 and k['private_segment_fixed_size'] == 0, (name, k)
    assert seen >= 40, seen


def test_persistent_kernels_keep_their_occupancy(kernels):
    """register counts that decide how many waves share a SIMD: the eight-wave bodies <= 256 (two waves), the four-wave bodies <= 512 (one),
    the attention forward <= 128 (four)"""
    for name, k in kernels.items():
        if 'attn_fwd_bf16_kernel' in name or 'attn_fwd_stream_kernel' in name:   # (the streamed forward: sixteen waves per workgroup = four per SIMD)
            assert k['vgpr_count'] <= 128, (name, k)
        elif 'attn_bwd_pers_kernel' in name or re.search(r'gemm_nt_kernelI', name) or re.search(r'gemm_wgrad8?_kernelI', name):
            assert k['vgpr_count'] <= 256, (name, k)
        elif 'kernel_4w' in name:
            assert k['vgpr_count'] <= 512, (name, k)
