"""ctypes binding of the TOOLS build of the library (tools/ecgvit_hip_tools.h; `make -C ecg-representation-learning_amd/csrc tools`):
the product C-ABI plus the test / diagnostic entry points.  Used by tests/hiputil.py and tools/*.py -- never by the product package."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TOOLS_LIB_PATH = os.path.join(ROOT, 'ecg-representation-learning_amd', 'csrc', 'build', 'libecgvit_hip_tools.so')
_tools = None


def tools_lib():
    global _tools
    if _tools is None:
        from ecg_representation_learning_amd import hip
        if not os.path.exists(TOOLS_LIB_PATH):
            raise hip.HipLibraryMissing(f'{TOOLS_LIB_PATH} not found: build it with `make -C ecg-representation-learning_amd/csrc tools` '
                                        f'(__graft_entry__.build() does)')
        l = ctypes.CDLL(TOOLS_LIB_PATH)
        for name, (res, args) in hip.SIGNATURES.items():      # the product entry points of the same build
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        P, I, F, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint64
        l.ecgvit_probe_mfma_layout.restype, l.ecgvit_probe_mfma_layout.argtypes = I, [P, P]
        l.ecgvit_attention_bwd_oneitem.restype, l.ecgvit_attention_bwd_oneitem.argtypes = I, [P, P, P, P, P, I, I, I, I, F, F, U, I, P]
        l.ecgvit_tools_attn_variant.restype, l.ecgvit_tools_attn_variant.argtypes = I, [I]
        l.ecgvit_tools_attn_fwd_variant.restype, l.ecgvit_tools_attn_fwd_variant.argtypes = I, [I]
        l.ecgvit_debug_attn_stamps.restype, l.ecgvit_debug_attn_stamps.argtypes = I, [P]
        _tools = l
    return _tools
