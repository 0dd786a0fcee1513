"""
ctypes binding of `libecgvit_hip.so` (C-ABI declared in `include/ecgvit_hip.h`).

Plumbing only: torch supplies device memory (`tensor.data_ptr()`) and the current HIP stream; every
numeric op is a hand-written gfx950 kernel behind the C-ABI.  There is NO fallback: if the library is
missing or a call returns non-zero this module raises -- the product path never degrades to eager torch.
"""
import ctypes
import os
from ctypes import c_int, c_int32, c_int64, c_uint64, c_float, c_void_p, c_char_p, POINTER, Structure, byref

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, 'libecgvit_hip.so')   # the ONE library a drop-in user loads; no environment variable changes it

ABI_VERSION = 6   # what ecgvit_abi_version() of a matching library returns (include/ecgvit_hip.h)
F32, BF16, FP8_E4M3, BF8_E5M2 = 0, 1, 2, 3
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_BIAS, EPI_GELU, EPI_GELU_BWD, EPI_RESIDUAL, EPI_ACCUM, EPI_DROPOUT, EPI_COLSUM = 1, 2, 4, 8, 16, 32, 64
EPI_GELU_GRAD_AUX, EPI_MUL_AUX, EPI_QUANT_OUT, EPI_NO_OUT, EPI_AUX8 = 128, 256, 512, 1024, 2048

KERNEL_NONE, KERNEL_GEMM_F32, KERNEL_GEMM_BF16, KERNEL_GEMM_NT, KERNEL_GEMM_WGRAD = 0, 1, 2, 3, 4

_ERR = {1: 'ECGVIT_EINVAL (unsupported shape / argument)', 2: 'ECGVIT_ELAUNCH (HIP launch failure)'}


class HipLibraryMissing(ImportError):
    pass


class GemmDesc(Structure):
    _fields_ = [
        ('layout', c_int32), ('dtype', c_int32), ('out_dtype', c_int32), ('epilogue', c_int32),
        ('M', c_int32), ('N', c_int32), ('K', c_int32), ('batch1', c_int32), ('batch2', c_int32),
        ('A', c_void_p), ('lda', c_int64), ('strideA1', c_int64), ('strideA2', c_int64),
        ('B', c_void_p), ('ldb', c_int64), ('strideB1', c_int64), ('strideB2', c_int64),
        ('C', c_void_p), ('ldc', c_int64), ('strideC1', c_int64), ('strideC2', c_int64),
        ('bias', c_void_p), ('residual', c_void_p), ('ldr', c_int64), ('aux', c_void_p), ('ldaux', c_int64),
        ('alpha', c_float), ('dropout_p', c_float), ('dropout_seed', c_uint64),
        ('workspace', c_void_p), ('workspace_bytes', c_int64), ('colsum_out', c_void_p), ('tiles_per_workgroup', c_int32),
        ('q8_out', c_void_p), ('ldq8', c_int64), ('q8_scale', c_void_p), ('q8_amax', c_void_p), ('q8_format', c_int32),
        ('scale_a', c_void_p), ('scale_b', c_void_p),
    ]


# name -> (restype, argtypes); mirrors include/ecgvit_hip.h one to one
_P, _I, _L, _F, _U = c_void_p, c_int, c_int64, c_float, c_uint64
SIGNATURES = {
    'ecgvit_version': (c_char_p, []),
    'ecgvit_abi_version': (c_int, []),
    'ecgvit_gemm': (c_int, [POINTER(GemmDesc), _P]),
    'ecgvit_gemm_workspace': (c_int64, [POINTER(GemmDesc)]),
    'ecgvit_gemm_kernel': (c_int, [POINTER(GemmDesc)]),
    'ecgvit_fp8_amax': (c_int, [_P, _P, _I, _L, _P, _P]),
    'ecgvit_fp8_quantize': (c_int, [_P, _P, _P, _I, _L, _I, _P, _P, _P]),
    'ecgvit_fp8_scale_update': (c_int, [_P, _P, _I, _P, _I, _P]),
    'ecgvit_patch_gather': (c_int, [_P, _P, _I, _I, _I, _I, _L, _I, _P]),
    'ecgvit_patch_gather_transform': (c_int, [_P, _P, _I, _I, _I, _I, _I, _L, _P, _P, _P, _P, _I, _P]),
    'ecgvit_embed_finish': (c_int, [_P, _P, _P, _P, _I, _I, _I, _F, _U, _I, _P]),
    'ecgvit_embed_bwd': (c_int, [_P, _P, _P, _P, _I, _I, _I, _F, _U, _I, _P]),
    'ecgvit_layernorm_fwd': (c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P]),
    'ecgvit_layernorm_fwd_q8': (c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _F, _P, _P, _P, _P]),
    'ecgvit_layernorm_bwd_workspace': (c_int64, [_L, _I]),
    'ecgvit_layernorm_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    'ecgvit_layernorm_bwd_fused': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _P, _F, _U, _I, _P]),
    'ecgvit_layernorm_bwd_fused_q8': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _P, _F, _U, _P, _P, _P, _P]),
    'ecgvit_dropout_apply': (c_int, [_P, _P, _L, _F, _U, _I, _P]),
    'ecgvit_colsum_workspace': (c_int64, [_L, _I]),
    'ecgvit_colsum': (c_int, [_P, _L, _P, _P, _L, _I, _I, _P]),
    'ecgvit_attention_fwd': (c_int, [_P, _P, _P, _I, _I, _I, _I, _F, _F, _U, _I, _P]),
    'ecgvit_attention_fwd_q8': (c_int, [_P, _P, _P, _I, _I, _I, _I, _F, _F, _U, _P, _P, _P, _P]),
    'ecgvit_attention_bwd': (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _U, _I, _P]),
    'ecgvit_attention_bwd_q8': (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _U, _P, _P, _P, _P]),
    'ecgvit_attention_probs': (c_int, [_P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    'ecgvit_softmax_rows': (c_int, [_P, _L, _I, _L, _P]),
    'ecgvit_softmax_bwd_rows': (c_int, [_P, _P, _L, _I, _L, _F, _P]),
    'ecgvit_head_fwd': (c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P]),
    'ecgvit_bce_fwd': (c_int, [_P, _P, _P, _P, _P, _L, _P]),
    'ecgvit_bce_bwd': (c_int, [_P, _P, _P, _P, _P, _F, _P, _L, _P]),
    'ecgvit_head_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'ecgvit_sumsq_workspace': (c_int64, [_L]),
    'ecgvit_sumsq': (c_int, [_P, _L, _P, _P, _P]),
    'ecgvit_adamw_step': (c_int, [_P, _P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _F, _F, _I, _I, _P, _P]),
    'ecgvit_clip_scale': (c_int, [_P, _L, _P, _F, _P, _P]),
    'ecgvit_cast_f32_to_bf16': (c_int, [_P, _P, _L, _P]),
    'ecgvit_cast_bf16_to_f32': (c_int, [_P, _P, _L, _P]),
    'ecgvit_transpose_bf16_batched': (c_int, [_P, _P, _P, _I, _L, _P]),
    'ecgvit_mask_embed_finish': (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'ecgvit_mask_embed_bwd': (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'ecgvit_gather_rows': (c_int, [_P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _P]),
    'ecgvit_scatter_rows': (c_int, [_P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _P]),
    'ecgvit_l1_loss_fwd_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _L, _I, _P]),
    'ecgvit_eval_counts': (c_int, [_P, _L, _P, _L, _L, _I, _I, _I, _P, _P]),
}

_lib = None


def use_library(path):
    """Measurement tools only (tools/*, bench.py --hip-lib): load ANOTHER build of the same ABI (the stamped diagnostic build, an A/B
    candidate) instead of the shipped library.  Explicit call, before the first kernel call of the process -- never ambient state."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError('use_library() must come before the first ecgvit call of the process')
    if not os.path.exists(path):
        raise HipLibraryMissing(f'{path} not found')
    LIB_PATH = os.path.abspath(path)


def lib():
    """Load the C-ABI library (once). Raises HipLibraryMissing if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(or `make -C {os.path.join(_PKG_DIR, "csrc")}`). There is no CPU / eager fallback.')
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError here = header/library mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what}: {_ERR.get(rc, rc)}')


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def code(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise ValueError(f'unsupported activation dtype {dtype}')


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('ecgvit HIP ops need device tensors (no CPU fallback exists)')


# ------------------------------------------------------------------------------------------------
# thin wrappers (argument marshalling only)
# ------------------------------------------------------------------------------------------------
def gemm_desc(layout, A, B, C, M, N, K, lda, ldb, ldc, *, epilogue=0, bias=None, residual=None, ldr=0, aux=None, ldaux=0,
              alpha=1.0, dropout_p=0.0, seed=0, batch=(1, 1), strideA=(0, 0), strideB=(0, 0), strideC=(0, 0), workspace=None,
              a_off=0, b_off=0, c_off=0, colsum_out=None, tiles_per_workgroup=0, fp8_format=None, scale_a=None, scale_b=None, q8_out=None, ldq8=0, q8_scale=None,
              q8_amax=None, q8_format=0):
    """Fill an `ecgvit_gemm_desc`. A/B/C are tensors (base pointers); *_off are ELEMENT offsets into them (head / q-k-v column slices).
    C may be None with EPI_NO_OUT (8-bit emitting FFN-wide products whose consumers read the 8-bit copy only): bf16 is then the output type
    the 8-bit copy is rounded through."""
    if C is None:
        if not epilogue & EPI_NO_OUT:
            raise ValueError('gemm: C is None without EPI_NO_OUT')
        _need_cuda(A, B)
    else:
        _need_cuda(A, B, C)
    d = GemmDesc()
    d.layout, d.dtype, d.out_dtype, d.epilogue = layout, (fp8_format if fp8_format is not None else code(A.dtype)), (BF16 if C is None else code(C.dtype)), epilogue
    d.scale_a, d.scale_b = ptr(scale_a), ptr(scale_b)   # device scalars of 8-bit operands (A, B are uint8 tensors then)
    d.q8_out, d.ldq8, d.q8_scale, d.q8_amax, d.q8_format = ptr(q8_out), ldq8, ptr(q8_scale), ptr(q8_amax), q8_format
    d.M, d.N, d.K, d.batch1, d.batch2 = M, N, K, batch[0], batch[1]
    d.A, d.lda, d.strideA1, d.strideA2 = A.data_ptr() + a_off * A.element_size(), lda, strideA[0], strideA[1]
    d.B, d.ldb, d.strideB1, d.strideB2 = B.data_ptr() + b_off * B.element_size(), ldb, strideB[0], strideB[1]
    d.C, d.ldc, d.strideC1, d.strideC2 = (None if C is None else C.data_ptr() + c_off * C.element_size()), ldc, strideC[0], strideC[1]
    d.bias, d.residual, d.ldr, d.aux, d.ldaux = ptr(bias), ptr(residual), ldr, ptr(aux), ldaux
    d.alpha, d.dropout_p, d.dropout_seed = alpha, dropout_p, seed
    d.colsum_out = ptr(colsum_out)
    d.tiles_per_workgroup = tiles_per_workgroup   # 0 = persistent static shares; the engine passes 2 while RCCL buckets overlap its backward
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    return d


def gemm(layout, A, B, C, M, N, K, lda, ldb, ldc, **kw):
    """C = epilogue(alpha * op(A) . op(B)); keyword arguments as `gemm_desc`."""
    d = gemm_desc(layout, A, B, C, M, N, K, lda, ldb, ldc, **kw)
    check(lib().ecgvit_gemm(byref(d), stream()), 'ecgvit_gemm')


def gemm_kernel(layout, A, B, C, M, N, K, lda, ldb, ldc, **kw):
    """KERNEL_* id of the kernel family `gemm()` would launch for these arguments (the library's own dispatch, nothing launched)"""
    d = gemm_desc(layout, A, B, C, M, N, K, lda, ldb, ldc, **kw)
    return lib().ecgvit_gemm_kernel(byref(d))


def gemm_workspace_bytes(layout, dtype, M, N, K):
    d = GemmDesc()
    d.layout, d.dtype, d.out_dtype, d.M, d.N, d.K, d.batch1, d.batch2 = layout, code(dtype), F32, M, N, K, 1, 1
    return lib().ecgvit_gemm_workspace(byref(d))
