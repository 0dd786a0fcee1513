"""-m gpu: the fp8 operand path (BASELINE.json configs[4]) -- quantise passes against torch's float8 casts, the 8-bit A.B^T kernel
against an f32 product of the dequantised operands, and the model-level fp8 Linear path against the bf16 path."""
import pytest
import torch

from hiputil import dev, rel_err
import ecg_representation_learning_amd as E
from ecg_representation_learning_amd import hip
from ecg_representation_learning_amd.hip import lib, check, ptr, stream

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
FMT = {hip.FP8_E4M3: (torch.float8_e4m3fn, 448.0), hip.BF8_E5M2: (torch.float8_e5m2, 57344.0)}


@pytest.mark.parametrize('fmt', [hip.FP8_E4M3, hip.BF8_E5M2])
def test_quantize_matches_torch_float8_cast(fmt):
    tdt, fmax = FMT[fmt]
    g = torch.Generator().manual_seed(3)
    n = 1 << 20
    x = (torch.randn(n, generator=g) * torch.exp(torch.randn(n, generator=g))).to(BF16).cuda()
    amax = torch.zeros(1, device='cuda')
    scale = torch.zeros(1, device='cuda')
    check(lib().ecgvit_fp8_amax(ptr(x), None, 1, n, ptr(amax), stream()), 'amax')
    assert float(amax) == float(x.float().abs().max())
    check(lib().ecgvit_fp8_scale_update(ptr(scale), ptr(amax), 1, None, fmt, stream()), 'scale_update')
    assert float(amax) == 0.0 and abs(float(scale) * fmax / float(x.float().abs().max()) - 1) < 1e-6
    y = torch.zeros(n, dtype=torch.uint8, device='cuda')
    nxt = torch.zeros(1, device='cuda')
    check(lib().ecgvit_fp8_quantize(ptr(x), ptr(y), None, 1, n, fmt, ptr(scale), ptr(nxt), stream()), 'quantize')
    assert float(nxt) == float(x.float().abs().max())
    want = (x.float() * (1.0 / scale)).clamp(-fmax, fmax).to(tdt)     # the kernel multiplies by the f32 reciprocal of the scale
    got = y.view(tdt)
    same = (got.view(torch.uint8) == want.view(torch.uint8)) | ((got.float() == 0) & (want.float() == 0))   # +0 / -0
    assert float(same.float().mean()) > 0.9999, float(same.float().mean())                                   # RNE both ways
    assert rel_err(got.float() * scale, x.float()) < (0.04 if fmt == hip.FP8_E4M3 else 0.08)                 # 3 / 2 mantissa bits
    # a stale (too small) scale saturates instead of overflowing to NaN / inf
    small = (scale * 0.25).clone()
    check(lib().ecgvit_fp8_quantize(ptr(x), ptr(y), None, 1, n, fmt, ptr(small), None, stream()), 'quantize')
    assert torch.isfinite(y.view(tdt).float()).all() and float(y.view(tdt).float().abs().max()) == fmax


def test_quantize_segments():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(10000 * 8, generator=g).to(BF16).cuda()
    table = torch.tensor([[0, 4096], [8192, 1024], [16384, 40000]], dtype=torch.int64, device='cuda')
    amax = torch.zeros(3, device='cuda')
    x[8192:8192 + 1024] *= 7
    check(lib().ecgvit_fp8_amax(ptr(x), ptr(table), 3, 40000, ptr(amax), stream()), 'amax')
    want = [float(x[o:o + c].float().abs().max()) for o, c in table.tolist()]
    assert amax.tolist() == want
    scale = torch.zeros(3, device='cuda')
    check(lib().ecgvit_fp8_scale_update(ptr(scale), ptr(amax), 3, None, hip.FP8_E4M3, stream()), 'scale_update')
    y = torch.full((x.numel(),), 0x7F, dtype=torch.uint8, device='cuda')
    check(lib().ecgvit_fp8_quantize(ptr(x), ptr(y), ptr(table), 3, 40000, hip.FP8_E4M3, ptr(scale), None, stream()), 'quantize')
    for i, (o, c) in enumerate(table.tolist()):
        assert rel_err(y[o:o + c].view(torch.float8_e4m3fn).float() * scale[i], x[o:o + c].float()) < 0.04
    assert bool((y[4096:8192] == 0x7F).all())          # bytes between segments untouched


def _rand8(shape, tdt, g, spread=1.0):
    return (torch.randn(*shape, generator=g) * spread).to(tdt)


@pytest.mark.parametrize('afmt', [hip.FP8_E4M3, hip.BF8_E5M2])
@pytest.mark.parametrize('shape', [(4133, 776, 384), (70011, 768, 1024), (2048, 3072, 768)])
def test_gemm_8bit_operands_vs_f32_product(afmt, shape):
    """products of 8-bit values are exact in f32 and the kernel accumulates in f32: C must equal the f32 product of the dequantised
    operands up to accumulation order and the bf16 rounding of the output; ragged M and N, scales through device scalars"""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + K)
    A = _rand8((M, K), FMT[afmt][0], g, 2.0).cuda()
    B = _rand8((N, K), torch.float8_e4m3fn, g, 0.5).cuda()
    sa, sb = torch.tensor([0.37], device='cuda'), torch.tensor([1.9], device='cuda')
    C = torch.full((M, N), float('nan'), device='cuda', dtype=BF16)
    hip.gemm(hip.GEMM_NT, A.view(torch.uint8), B.view(torch.uint8), C, M, N, K, K, K, N, fp8_format=afmt, scale_a=sa, scale_b=sb)
    ref = (A.float() @ B.float().t()) * 0.37 * 1.9
    assert torch.isfinite(C.float()).all()
    assert rel_err(C, ref) < 3e-3
    # small integers: every product and partial sum exact -> equality after the bf16 rounding, on repeated launches
    Ai = torch.randint(-2, 3, (M, K), generator=g).float().to(FMT[afmt][0]).cuda()
    Bi = torch.randint(-2, 3, (N, K), generator=g).float().to(torch.float8_e4m3fn).cuda()
    one = torch.ones(1, device='cuda')
    want = (Ai.float() @ Bi.float().t()).to(BF16)
    for _ in range(3):
        C.fill_(float('nan'))
        hip.gemm(hip.GEMM_NT, Ai.view(torch.uint8), Bi.view(torch.uint8), C, M, N, K, K, K, N, fp8_format=afmt, scale_a=one, scale_b=one)
        assert torch.equal(C, want)


def test_gemm_8bit_epilogues():
    M, N, K = 5000, 1024, 512
    g = torch.Generator().manual_seed(9)
    A = _rand8((M, K), torch.float8_e4m3fn, g).cuda()
    B = _rand8((N, K), torch.float8_e4m3fn, g, 0.2).cuda()
    s = torch.tensor([0.5], device='cuda')
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(BF16).cuda()
    C = torch.empty(M, N, device='cuda', dtype=BF16)
    hip.gemm(hip.GEMM_NT, A.view(torch.uint8), B.view(torch.uint8), C, M, N, K, K, K, N, fp8_format=hip.FP8_E4M3, scale_a=s, scale_b=s,
             epilogue=hip.EPI_BIAS | hip.EPI_RESIDUAL, bias=bias, residual=res, ldr=N)
    ref = (A.float() @ B.float().t()) * 0.25 + bias + res.float()
    assert rel_err(C, ref) < 4e-3
    aux = torch.empty(M, N, device='cuda', dtype=BF16)
    hip.gemm(hip.GEMM_NT, A.view(torch.uint8), B.view(torch.uint8), C, M, N, K, K, K, N, fp8_format=hip.FP8_E4M3, scale_a=s, scale_b=s,
             epilogue=hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_GELU_GRAD_AUX, bias=bias, aux=aux, ldaux=N)
    pre = (A.float() @ B.float().t()) * 0.25 + bias
    assert rel_err(C, torch.nn.functional.gelu(pre)) < 4e-3
