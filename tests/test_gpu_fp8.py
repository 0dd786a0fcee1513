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


def test_model_fp8_linear_vs_bf16_path():
    """EcgVit-base layer shape, 2 layers, 12 x 5000 / patch 20: the fp8 Linear path against the bf16 path on the same weights and batch
    (dropout 0).  Tolerances: e4m3 has 3 mantissa bits (2^-4 relative per element, averaged down by the K = 768..3072 sums): loss within
    3 % relative, total gradient cosine >= 0.97, every tensor's >= 0.90; then three fused steps stay finite and reduce the loss."""
    from oracle import vit_oracle as O
    conf = E.EcgVitConfig(max_signal_length=5000, patch_size=20, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072,
                          hidden_dropout_prob=0., attention_probs_dropout_prob=0.)
    torch.manual_seed(5)
    m16 = E.EcgVit(config=conf, compute_dtype=BF16).cuda().train()
    m8 = E.EcgVit(config=conf, compute_dtype=BF16, fp8_linear=True)
    m8.load_state_dict(m16.state_dict())
    m8.cuda().train()
    x, y = O.synthetic_batch(12, length=5000, seed=5)     # 12 x 251 = 3012 token rows: above the 8-bit kernel's 2048-row floor
    x, y = x.cuda(), y.cuda()
    o16 = m16(sample_values=x, labels=y)
    o8 = m8(sample_values=x, labels=y)
    assert len(m8._engine()._f8_seen) == 8, 'the 8-bit Linear path did not run'    # 4 forward sites x 2 layers
    assert torch.isfinite(o8.logits).all()
    assert abs(float(o8.loss) - float(o16.loss)) / float(o16.loss) < 3e-2
    o16.loss.backward(); o8.loss.backward()
    assert len(m8._engine()._f8_seen) == 16                                             # + 4 gradient sites x 2 layers
    g16 = torch.cat([p.grad.flatten() for p in m16.parameters()]).double()
    g8 = torch.cat([p.grad.flatten() for p in m8.parameters()]).double()
    assert torch.isfinite(g8).all()
    cos = float((g16 @ g8) / (g16.norm() * g8.norm()))
    assert cos > 0.97, cos
    for (k, p), (_, q) in zip(m8.named_parameters(), m16.named_parameters()):
        c = float((p.grad.double().flatten() @ q.grad.double().flatten()) / (p.grad.double().norm() * q.grad.double().norm() + 1e-30))
        assert c > 0.90, (k, c)
    # second forward: delayed scales (from the first pass's amax) instead of the first-use amax pass -- same result to fp8 noise
    o8b = m8(sample_values=x, labels=y)
    assert abs(float(o8b.loss) - float(o8.loss)) / float(o8.loss) < 1e-2
    step = E.HipTrainStep(m8, E.get_train_args(dict(train_batch_size=12, num_train_epoch=1, warmup_ratio=0.0), n_train=12 * 20))
    losses = [float(step.step(x, y)[0]) for _ in range(4)]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses


def test_fp8_large_layer_shape_vs_cpu_oracle():
    """The fp8 Linear path against the CPU ORACLE (torch eager f32 restatement of the reference step), not against another HIP path:
    EcgVit-large layer shape (d = 1024, 16 heads, ffn 4096), patch 10 -> 501 tokens, 2 layers, 9 records (4 509 token rows: above the
    8-bit products' 2 048-row floor and the 8-bit weight-gradient kernel's 4 096-row floor), dropout 0.  Tolerances (written here, BASELINE.json configs[4]): e4m3 operands carry 3 mantissa
    bits (2^-4 relative per element, averaged down by the K = 1024 .. 4096 sums), e5m2 gradients 2 bits; the loss must sit within 3 %
    of the oracle's, the logits within 0.2 absolute, the whole gradient at cosine >= 0.97 and every parameter tensor at >= 0.90."""
    from oracle import vit_oracle as O
    torch.set_num_threads(min(32, torch.get_num_threads()))
    conf = E.EcgVitConfig(max_signal_length=5000, patch_size=10, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096,
                          hidden_dropout_prob=0., attention_probs_dropout_prob=0.)
    torch.manual_seed(77)
    ref = O.OracleEcgVit(config=conf).train()
    m8 = E.EcgVit(config=conf, compute_dtype=BF16, fp8_linear=True)
    m8.load_state_dict(ref.state_dict())
    m8.cuda().train()
    x, y = O.synthetic_batch(9, length=5000, seed=77)
    o_ref = ref(sample_values=x, labels=y)
    o_ref.loss.backward()
    kinds, real = [], hip.gemm

    def spy(layout, *a, **k):
        kinds.append((layout, k.get('fp8_format')))
        return real(layout, *a, **k)
    hip.gemm = spy
    try:
        o8 = m8(sample_values=x.cuda(), labels=y.cuda())
        o8.loss.backward()
    finally:
        hip.gemm = real
    assert len(m8._engine()._f8_seen) == 16, 'the 8-bit Linear path did not run'         # 8 sites x 2 layers
    assert kinds.count((hip.GEMM_TN, hip.BF8_E5M2)) == 8, kinds                           # the 4 weight gradients of both layers ran on 8-bit operands

    def hold_against_oracle(o, tag):
        lref = float(o_ref.loss.detach())
        assert abs(float(o.loss.detach()) - lref) / lref < 3e-2, (tag, float(o.loss.detach()), lref)
        assert float((o.logits.detach().cpu() - o_ref.logits.detach()).abs().max()) < 0.2, tag
        gref = torch.cat([p.grad.flatten() for p in ref.parameters()]).double()
        g8 = torch.cat([p.grad.flatten() for p in m8.parameters()]).double().cpu()
        assert torch.isfinite(g8).all(), tag
        cos = float((g8 @ gref) / (g8.norm() * gref.norm()))
        assert cos > 0.97, (tag, cos)
        for (k, p), (_, q) in zip(m8.named_parameters(), ref.named_parameters()):
            c = float((p.grad.double().cpu().flatten() @ q.grad.double().flatten()) / (p.grad.double().norm().cpu() * q.grad.double().norm() + 1e-30))
            assert c > 0.90, (tag, k, c)
    hold_against_oracle(o8, 'first pass (first-use scales, quantise passes)')

    # STEADY STATE -- what bench.py times from its second step on: the scales are DELAYED (from the previous pass's amax) and the producers emit
    # the 8-bit operand copies themselves (LayerNorm forward / backward, the attention kernels, EPI_QUANT_OUT epilogues).  Same weights, same
    # batch, same oracle, same tolerances; the emitting entry points must actually have run.
    eng = m8._engine()
    fired = {'epi_quant_out': 0, 'epi_aux8': 0, 'layernorm_fwd_q8': 0, 'layernorm_bwd_fused_q8': 0, 'attention_fwd_q8': 0, 'attention_bwd_q8': 0, 'scale_update': 0}
    first_use = {'fp8_amax': 0}
    l = hip.lib()

    def counting(name, key):
        fn = getattr(l, name)

        def wrapped(*a):
            fired[key] += 1
            return fn(*a)
        return fn, wrapped
    saved = {}
    for name, key in (('ecgvit_layernorm_fwd_q8', 'layernorm_fwd_q8'), ('ecgvit_layernorm_bwd_fused_q8', 'layernorm_bwd_fused_q8'),
                      ('ecgvit_attention_fwd_q8', 'attention_fwd_q8'), ('ecgvit_attention_bwd_q8', 'attention_bwd_q8'),
                      ('ecgvit_fp8_scale_update', 'scale_update')):
        saved[name], w = counting(name, key)
        setattr(l, name, w)
    amax_fn = l.ecgvit_fp8_amax
    saved['ecgvit_fp8_amax'] = amax_fn

    def amax_spy(*a):
        first_use['fp8_amax'] += 1
        return amax_fn(*a)
    l.ecgvit_fp8_amax = amax_spy

    def spy2(layout, *a, **k):
        if k.get('epilogue', 0) & hip.EPI_QUANT_OUT:
            fired['epi_quant_out'] += 1
        if k.get('epilogue', 0) & hip.EPI_AUX8:          # the saved FFN tensor as e4m3 bytes (round 5): both FFN-wide launches of both layers
            fired['epi_aux8'] += 1
        return real(layout, *a, **k)
    hip.gemm = spy2
    try:
        for p in m8.parameters():
            p.grad = None
        o8s = m8(sample_values=x.cuda(), labels=y.cuda())
        o8s.loss.backward()
        torch.cuda.synchronize()
    finally:
        hip.gemm = real
        for name, fn in saved.items():
            setattr(l, name, fn)
    assert all(v > 0 for v in fired.values()), fired          # the delayed scale update and every emitting producer ran in this pass
    assert first_use['fp8_amax'] == 0, first_use              # ... and no site took the first-use (own amax) path again
    hold_against_oracle(o8s, 'steady state (delayed scales, producers emit the 8-bit copies)')
    # EVAL passes follow the data too (advisor, round 4): an inference-only model must not keep its first batch's scales.  The first block's
    # LayerNorm gain x 4 makes the QKV input (site 0) four times as large as anything the scales have seen: the pass after the first one on it
    # runs with the scale grown to the new amax (with the old scale everything above the old amax was clamped), and its logits agree with
    # the bf16 path's
    m16 = E.EcgVit(config=conf, compute_dtype=BF16)
    m16.load_state_dict(ref.state_dict())
    m16.cuda().eval()
    m8.eval()
    for mm in (m8, m16):
        dict(mm.named_parameters())['vit.transformer.layers.0.0.norm.weight'].data.mul_(4.0)
    sc = eng.f8_scale.clone()
    with torch.no_grad():
        m8(sample_values=x.cuda())                       # scales still from the pass before; its amax is recorded
        l8 = m8(sample_values=x.cuda()).logits.clone()   # delayed scales from the pass above
        l16 = m16(sample_values=x.cuda()).logits.clone()
    assert 3.0 < float(eng.f8_scale[0]) / float(sc[0]) < 5.0, (float(eng.f8_scale[0]), float(sc[0]))
    assert float((l8 - l16).abs().max()) < 0.2, float((l8 - l16).abs().max())
    dict(m8.named_parameters())['vit.transformer.layers.0.0.norm.weight'].data.mul_(0.25)
    m8.train()


def test_full_large_fp8_configuration_properties():
    """BASELINE.json configs[4] on one GPU as benchmarked: EcgVit-large, patch 10 (501 tokens), fp8 Linear operands, dropout 0.1,
    256 records per GPU: finite, bit-identical rerun under a pinned seed, a falling loss over three fused steps"""
    conf = E.EcgVitConfig.from_defined('ecg-vit-large')
    conf.max_signal_length, conf.patch_size = 5000, 10
    torch.manual_seed(77)
    m = E.EcgVit(config=conf, compute_dtype=BF16, fp8_linear=True).cuda().train()
    x, y = E.workload.synthetic_batch(256, length=5000, seed=77)
    x, y = x.cuda(), y.cuda()
    eng = m._engine()
    eng.forward(x, y, None, training=True, seed=1, want_mean=True)                 # settles the first-use scales
    la, _, ma = (t.clone() for t in eng.forward(x, y, None, training=True, seed=4711, want_mean=True))
    eng.f8_amax.zero_()                                                            # same scales for the rerun
    lb, _, mb = (t.clone() for t in eng.forward(x, y, None, training=True, seed=4711, want_mean=True))
    assert torch.isfinite(la).all() and torch.equal(la, lb) and torch.equal(ma, mb)
    step = E.HipTrainStep(m, E.get_train_args(dict(train_batch_size=256, num_train_epoch=1, warmup_ratio=0.0), n_train=256 * 20))
    losses = [float(step.step(x, y)[0]) for _ in range(3)]
    step.finish()
    assert all(l == l and abs(l) < 1e4 for l in losses) and losses[2] < losses[0], losses


@pytest.mark.parametrize('p', [0.0, 0.1])
def test_layernorm_bwd_emits_the_e5m2_copy_of_its_gradient(p):
    """ecgvit_layernorm_bwd_fused_q8 = ecgvit_layernorm_bwd_fused bit for bit (dx, dxm, dgamma, dbeta, column sums) + the e5m2 copy of the
    gradient the next stage consumes (dxm under dropout, dx without), equal to torch's float8_e5m2 cast of that tensor over the scale, and
    its amax"""
    rows, d = 4100, 1024
    g = torch.Generator().manual_seed(21)
    dy, x, dres = (torch.randn(rows, d, generator=g).to(BF16).cuda() for _ in range(3))
    gamma = torch.randn(d, generator=g).cuda()
    mean, rstd = x.float().mean(1), 1.0 / torch.sqrt(x.float().var(1, unbiased=False) + 1e-5)
    ws = torch.empty(lib().ecgvit_layernorm_bwd_workspace(rows, d), dtype=torch.uint8, device='cuda')

    def run(q8):
        dx, dxm = torch.zeros(rows, d, device='cuda', dtype=BF16), torch.zeros(rows, d, device='cuda', dtype=BF16)
        dg, db, cs = torch.zeros(d, device='cuda'), torch.zeros(d, device='cuda'), torch.zeros(d, device='cuda')
        if q8 is None:
            check(lib().ecgvit_layernorm_bwd_fused(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dg), ptr(db), ptr(ws), rows, d,
                                                   ptr(dxm), ptr(cs), p, 77, hip.BF16, stream()), 'ln_bwd_fused')
        else:
            g8, scale, amax = q8
            check(lib().ecgvit_layernorm_bwd_fused_q8(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dg), ptr(db), ptr(ws), rows, d,
                                                      ptr(dxm), ptr(cs), p, 77, ptr(g8), ptr(scale), ptr(amax), stream()), 'ln_bwd_fused_q8')
        return dx, dxm, dg, db, cs
    ref = run(None)
    grad = ref[1] if p > 0 else ref[0]
    scale = (grad.float().abs().max() / 57344.0 * 1.5).reshape(1)        # a delayed scale: not exactly this tensor's
    g8 = torch.full((rows, d), 0x7F, dtype=torch.uint8, device='cuda')
    amax = torch.zeros(1, device='cuda')
    got = run((g8, scale, amax))
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    assert float(amax) == float(grad.float().abs().max())
    want = (grad.float() * (1.0 / scale)).clamp(-57344.0, 57344.0).to(torch.float8_e5m2)
    same = (g8.view(torch.float8_e5m2).view(torch.uint8) == want.view(torch.uint8)) | ((g8.view(torch.float8_e5m2).float() == 0) & (want.float() == 0))
    assert float(same.float().mean()) > 0.9999


@pytest.mark.parametrize('afmt', [hip.BF8_E5M2, hip.FP8_E4M3])
@pytest.mark.parametrize('shape', [(512, 256, 70011), (1024, 3072, 16384), (256, 512, 4100)])
def test_weight_gradient_8bit_operands_vs_f32_product(afmt, shape):
    """dW = dY8^T . X8 on the 8-bit streaming split-K kernel (ds_read_b64_tr_b8 fragments, block-scaled 32x32x64 MFMA, unit block scales):
    products of 8-bit values are exact in f32, so the result equals the f32 product of the dequantised operands up to summation order;
    ragged K (token rows, not a multiple of 128: rows beyond the end read as zero), scales through device scalars; small integers exactly,
    on repeated launches (race screen)"""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    A = _rand8((K, M), FMT[afmt][0], g, 2.0).cuda()
    B = _rand8((K, N), torch.float8_e4m3fn, g, 0.5).cuda()
    sa, sb = torch.tensor([0.37], device='cuda'), torch.tensor([1.9], device='cuda')
    d = hip.gemm_desc(hip.GEMM_TN, A.view(torch.uint8), B.view(torch.uint8), torch.empty(M, N, device='cuda'), M, N, K, M, N, N, fp8_format=afmt)
    ws = torch.empty(max(lib().ecgvit_gemm_workspace(hip.byref(d)), 16), dtype=torch.uint8, device='cuda')
    assert hip.gemm_kernel(hip.GEMM_TN, A.view(torch.uint8), B.view(torch.uint8), torch.empty(M, N, device='cuda'), M, N, K, M, N, N, fp8_format=afmt,
                           workspace=ws) == hip.KERNEL_GEMM_WGRAD
    C = torch.full((M, N), float('nan'), device='cuda')
    hip.gemm(hip.GEMM_TN, A.view(torch.uint8), B.view(torch.uint8), C, M, N, K, M, N, N, fp8_format=afmt, scale_a=sa, scale_b=sb, workspace=ws)
    ref = (A.float().t() @ B.float()) * (0.37 * 1.9)
    assert torch.isfinite(C).all()
    assert rel_err(C, ref) < 5e-5      # f32 summation order over up to 70 011 terms (the torch reference rounds too); exactness: below
    Ai = torch.randint(-2, 3, (K, M), generator=g).float().to(FMT[afmt][0]).cuda()
    Bi = torch.randint(-2, 3, (K, N), generator=g).float().to(torch.float8_e4m3fn).cuda()
    one = torch.ones(1, device='cuda')
    want = Ai.float().t() @ Bi.float()
    for _ in range(3):
        C.fill_(float('nan'))
        hip.gemm(hip.GEMM_TN, Ai.view(torch.uint8), Bi.view(torch.uint8), C, M, N, K, M, N, N, fp8_format=afmt, scale_a=one, scale_b=one, workspace=ws)
        assert torch.equal(C, want)


@pytest.mark.parametrize('N,p,B', [(251, 0.1, 20), (501, 0.0, 20), (200, 0.1, 20), (251, 0.1, 140), (501, 0.1, 70), (501, 0.0, 65), (251, 0.0, 129)])   # B >= 129 / 65: the STREAMED forward (a workgroup's worth of items per CU)
def test_attention_kernels_emit_their_8bit_copies(N, p, B):
    """ecgvit_attention_fwd_q8 / _bwd_q8 = the plain kernels bit for bit (out, lse, dqkv) + the e4m3 copy of `out` / the e5m2 copy of
    `dqkv`, equal to torch's float8 casts of those tensors over the given scales, and their amax (one and two key windows)"""
    h, dh = 4, 64
    d = h * dh
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn(B * N, 3 * d, generator=g).to(BF16).cuda()
    do = torch.randn(B * N, d, generator=g).to(BF16).cuda()
    out, out2 = (torch.zeros(B * N, d, device='cuda', dtype=BF16) for _ in range(2))
    lse, lse2 = (torch.zeros(B * h * N, device='cuda') for _ in range(2))
    check(lib().ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, dh, 0.125, p, 7, hip.BF16, stream()), 'fwd')
    s_out = (out.float().abs().max() / 448.0 * 1.3).reshape(1)
    out8 = torch.full((B * N, d), 0x7F, dtype=torch.uint8, device='cuda')
    amax = torch.zeros(1, device='cuda')
    check(lib().ecgvit_attention_fwd_q8(ptr(qkv), ptr(out2), ptr(lse2), B, N, h, dh, 0.125, p, 7, ptr(out8), ptr(s_out), ptr(amax), stream()), 'fwd_q8')
    assert torch.equal(out, out2) and torch.equal(lse, lse2)
    assert float(amax) == float(out.float().abs().max())
    want = (out.float() * (1.0 / s_out)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    got = out8.view(torch.float8_e4m3fn)
    same = (got.view(torch.uint8) == want.view(torch.uint8)) | ((got.float() == 0) & (want.float() == 0))
    assert float(same.float().mean()) > 0.9999
    dq, dq2 = (torch.zeros(B * N, 3 * d, device='cuda', dtype=BF16) for _ in range(2))
    check(lib().ecgvit_attention_bwd(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(dq), B, N, h, dh, 0.125, p, 7, hip.BF16, stream()), 'bwd')
    s_dq = (dq.float().abs().max() / 57344.0 * 1.3).reshape(1)
    dq8 = torch.full((B * N, 3 * d), 0x7F, dtype=torch.uint8, device='cuda')
    amax.zero_()
    rc = lib().ecgvit_attention_bwd_q8(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(dq2), B, N, h, dh, 0.125, p, 7, ptr(dq8), ptr(s_dq), ptr(amax), stream())
    assert rc == 0
    assert torch.equal(dq, dq2)
    assert float(amax) == float(dq.float().abs().max())
    want = (dq.float() * (1.0 / s_dq)).clamp(-57344.0, 57344.0).to(torch.float8_e5m2)
    got = dq8.view(torch.float8_e5m2)
    same = (got.view(torch.uint8) == want.view(torch.uint8)) | ((got.float() == 0) & (want.float() == 0))
    assert float(same.float().mean()) > 0.9999, float(same.float().mean())
    # short sequences run the one-item kernel, which does not emit: the entry point says so instead of leaving the copy unwritten
    assert lib().ecgvit_attention_bwd_q8(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(dq2), 2, 100, h, dh, 0.125, p, 7, ptr(dq8), ptr(s_dq), ptr(amax), stream()) == 1
    # the 8-bit copy of the forward leaves in 16-byte stores: a copy that is not 16-byte aligned is refused, not written misaligned
    assert lib().ecgvit_attention_fwd_q8(ptr(qkv), ptr(out2), ptr(lse2), B, N, h, dh, 0.125, p, 7, out8.data_ptr() + 4, ptr(s_out), ptr(amax), stream()) == 1


def test_no_output_forms_emit_exactly_what_the_writing_forms_emit():
    """ABI 5: ECGVIT_EPI_NO_OUT on the two FFN-wide emitting bodies of the 8-bit A.B^T kernel, y == NULL in ecgvit_layernorm_fwd_q8 and
    dxm == NULL in ecgvit_layernorm_bwd_fused_q8 leave one bf16 tensor unwritten and change nothing else: the 8-bit copy, its amax, the saved
    GELU' x mask tensor, the column sums, dx, mean / rstd are bit for bit those of the writing call.  Other epilogues refuse the flag."""
    M, N, K = 5003, 1024, 512      # ragged last row tile
    g = torch.Generator().manual_seed(31)
    A = _rand8((M, K), torch.float8_e4m3fn, g).cuda()
    Bw = _rand8((N, K), torch.float8_e4m3fn, g, 0.2).cuda()
    s = torch.tensor([0.5], device='cuda')
    bias = torch.randn(N, generator=g).cuda()
    qs = torch.tensor([0.01], device='cuda')
    UP = hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_GELU_GRAD_AUX | hip.EPI_QUANT_OUT

    def up(no_out, drop):
        C = torch.full((M, N), 7.0, device='cuda', dtype=BF16)
        aux = torch.zeros(M, N, device='cuda', dtype=BF16)
        q8 = torch.full((M, N), 0x7F, dtype=torch.uint8, device='cuda')
        am = torch.zeros(1, device='cuda')
        hip.gemm(hip.GEMM_NT, A.view(torch.uint8), Bw.view(torch.uint8), None if no_out else C, M, N, K, K, K, N, fp8_format=hip.FP8_E4M3, scale_a=s, scale_b=s,
                 epilogue=UP | (hip.EPI_DROPOUT if drop else 0) | (hip.EPI_NO_OUT if no_out else 0), bias=bias, aux=aux, ldaux=N,
                 dropout_p=0.1 if drop else 0.0, seed=99, q8_out=q8, ldq8=N, q8_scale=qs, q8_amax=am, q8_format=hip.FP8_E4M3)
        torch.cuda.synchronize()
        return C, aux, q8, am
    for drop in (False, True):
        C0, aux0, q0, am0 = up(False, drop)
        C1, aux1, q1, am1 = up(True, drop)
        assert torch.equal(aux0, aux1) and torch.equal(q0, q1) and torch.equal(am0, am1) and float(am0) > 0
        assert bool((q0 != 0x7F).any()) and not bool((C0 == 7.0).all())
    # the FFN-down input gradient's body: x saved tensor + column sums, e5m2 gradients x e4m3 weights, e5m2 copy
    G8 = _rand8((M, K), torch.float8_e5m2, g).cuda()
    auxin = (torch.rand(M, N, generator=g) * 1.2).to(BF16).cuda()
    ws = torch.empty(max(lib().ecgvit_colsum_workspace(M, N), 8 * ((M + 255) // 256) * N), dtype=torch.uint8, device='cuda')
    DH = hip.EPI_MUL_AUX | hip.EPI_COLSUM | hip.EPI_QUANT_OUT

    def dh(no_out):
        C = torch.full((M, N), 7.0, device='cuda', dtype=BF16)
        q8 = torch.full((M, N), 0x7F, dtype=torch.uint8, device='cuda')
        am, cs = torch.zeros(1, device='cuda'), torch.zeros(N, device='cuda')
        hip.gemm(hip.GEMM_NT, G8.view(torch.uint8), Bw.view(torch.uint8), None if no_out else C, M, N, K, K, K, N, fp8_format=hip.BF8_E5M2, scale_a=s, scale_b=s,
                 epilogue=DH | (hip.EPI_NO_OUT if no_out else 0), aux=auxin, ldaux=N, workspace=ws, colsum_out=cs,
                 q8_out=q8, ldq8=N, q8_scale=qs, q8_amax=am, q8_format=hip.BF8_E5M2)
        torch.cuda.synchronize()
        return C, q8, am, cs
    C0, q0, am0, cs0 = dh(False)
    C1, q1, am1, cs1 = dh(True)
    assert torch.equal(q0, q1) and torch.equal(am0, am1) and torch.equal(cs0, cs1) and float(cs0.abs().max()) > 0
    # refused where no no-output body exists: without the 8-bit copy, on another epilogue, on bf16 operands
    Cx = torch.empty(M, N, device='cuda', dtype=BF16)
    for kw in (dict(epilogue=hip.EPI_NO_OUT), dict(epilogue=hip.EPI_BIAS | hip.EPI_QUANT_OUT | hip.EPI_NO_OUT, bias=bias, q8_out=q0, ldq8=N, q8_scale=qs, q8_amax=am0,
                                                   q8_format=hip.FP8_E4M3)):
        with pytest.raises(RuntimeError):
            hip.gemm(hip.GEMM_NT, A.view(torch.uint8), Bw.view(torch.uint8), Cx, M, N, K, K, K, N, fp8_format=hip.FP8_E4M3, scale_a=s, scale_b=s, **kw)
    Ab, Bb = torch.randn(M, K, generator=g).to(BF16).cuda(), torch.randn(N, K, generator=g).to(BF16).cuda()
    with pytest.raises(RuntimeError):
        hip.gemm(hip.GEMM_NT, Ab, Bb, Cx, M, N, K, K, K, N, epilogue=UP | hip.EPI_NO_OUT, bias=bias, aux=auxin, ldaux=N, q8_out=q0, ldq8=N, q8_scale=qs, q8_amax=am0,
                 q8_format=hip.FP8_E4M3)

    # LayerNorm forward: y == NULL
    rows, d = 4100, 1024
    x = torch.randn(rows, d, generator=g).to(BF16).cuda()
    gamma, beta = torch.randn(d, generator=g).cuda(), torch.randn(d, generator=g).cuda()
    sc = torch.tensor([0.02], device='cuda')

    def lnf(with_y):
        y = torch.full((rows, d), 7.0, device='cuda', dtype=BF16)
        y8 = torch.full((rows, d), 0x7F, dtype=torch.uint8, device='cuda')
        mean, rstd, am = torch.zeros(rows, device='cuda'), torch.zeros(rows, device='cuda'), torch.zeros(1, device='cuda')
        check(lib().ecgvit_layernorm_fwd_q8(ptr(x), ptr(gamma), ptr(beta), ptr(y) if with_y else None, ptr(mean), ptr(rstd), rows, d, 1e-5, ptr(y8), ptr(sc), ptr(am),
                                            stream()), 'ln_fwd_q8')
        torch.cuda.synchronize()
        return y, y8, mean, rstd, am
    a, b = lnf(True), lnf(False)
    assert all(torch.equal(u, v) for u, v in zip(a[1:], b[1:])) and bool((b[0] == 7.0).all()) and not bool((a[0] == 7.0).all())

    # fused LayerNorm backward: dxm == NULL under dropout
    dy, dres = (torch.randn(rows, d, generator=g).to(BF16).cuda() for _ in range(2))
    mean, rstd = x.float().mean(1), 1.0 / torch.sqrt(x.float().var(1, unbiased=False) + 1e-5)
    ws2 = torch.empty(lib().ecgvit_layernorm_bwd_workspace(rows, d), dtype=torch.uint8, device='cuda')
    sc2 = torch.tensor([1e-4], device='cuda')

    def lnb(with_dxm):
        dx, dxm = torch.zeros(rows, d, device='cuda', dtype=BF16), torch.full((rows, d), 7.0, device='cuda', dtype=BF16)
        dg, db, cs, am = torch.zeros(d, device='cuda'), torch.zeros(d, device='cuda'), torch.zeros(d, device='cuda'), torch.zeros(1, device='cuda')
        g8 = torch.full((rows, d), 0x7F, dtype=torch.uint8, device='cuda')
        check(lib().ecgvit_layernorm_bwd_fused_q8(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dg), ptr(db), ptr(ws2), rows, d,
                                                  ptr(dxm) if with_dxm else None, ptr(cs), 0.1, 77, ptr(g8), ptr(sc2), ptr(am), stream()), 'ln_bwd_fused_q8')
        torch.cuda.synchronize()
        return dxm, dx, dg, db, cs, g8, am
    a, b = lnb(True), lnb(False)
    assert all(torch.equal(u, v) for u, v in zip(a[1:], b[1:])) and bool((b[0] == 7.0).all()) and not bool((a[0] == 7.0).all())
    # (the non-emitting entry point still needs dxm under dropout)
    assert lib().ecgvit_layernorm_bwd_fused(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(a[1]), ptr(a[2]), ptr(a[3]), ptr(ws2), rows, d,
                                            None, ptr(a[4]), 0.1, 77, hip.BF16, stream()) != 0


def test_dropping_the_dead_bf16_tensors_changes_no_gradient():
    """fp8_linear steady state: with the 8-bit copies emitted by their producers, the bf16 xn1 / xn2 / hact / dh / dxm have 8-bit readers only and are
    not written (engine.fp8_drop_dead_bf16).  Three train steps with and without: losses and every parameter bit-identical."""
    from oracle import vit_oracle as O
    conf = E.EcgVitConfig(max_signal_length=5000, patch_size=20, hidden_size=512, num_hidden_layers=2, num_attention_heads=8, intermediate_size=1024,
                          hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    x, y = O.synthetic_batch(20, length=5000, seed=8)      # 20 x 251 = 5020 token rows: above the 8-bit weight-gradient kernel's 4096-row floor
    x, y = x.cuda(), y.cuda()
    outs = []
    for drop in (True, False):
        torch.manual_seed(6)
        m = E.EcgVit(config=conf, compute_dtype=BF16, fp8_linear=True).cuda().train()
        eng = m._engine()
        assert eng.fp8_drop_dead_bf16
        eng.fp8_drop_dead_bf16 = drop
        step = E.HipTrainStep(m, E.get_train_args(dict(train_batch_size=20, num_train_epoch=1, warmup_ratio=0.0), n_train=20 * 20))
        losses = [float(step.step(x, y)[0]) for _ in range(3)]
        if drop:   # the tensors really stayed unwritten in the last (steady-state) step: poison them, step, look
            for L in eng.act['layers']:
                for k in ('xn1', 'xn2', 'hact'):
                    L[k].fill_(7.0)
            eng.act['dh'].fill_(7.0)
            losses.append(float(step.step(x, y)[0]))
            for L in eng.act['layers']:
                for k in ('xn1', 'xn2', 'hact'):
                    assert bool((L[k] == 7.0).all()), k
            assert bool((eng.act['dh'] == 7.0).all())
        else:
            losses.append(float(step.step(x, y)[0]))
        outs.append((losses, torch.cat([p.detach().flatten() for p in m.parameters()]).clone()))
    assert outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


def test_training_scales_survive_an_eval_pass():
    """delayed scaling across train -> eval -> train (an evaluation between epochs): the eval pass may take its scales from the training pass
    before it (dropout-rescaled activations: larger, safe), but the training pass AFTER it must not take its forward scales from the eval pass --
    eval activations carry no 1 / (1 - p) rescale, scales have no headroom, and the FFN hidden activation would saturate in e4m3 for that step.
    The engine keeps the last training pass's scales there (and discards the eval pass's amax)."""
    from oracle import vit_oracle as O
    conf = E.EcgVitConfig(max_signal_length=5000, patch_size=20, hidden_size=512, num_hidden_layers=2, num_attention_heads=8, intermediate_size=2048,
                          hidden_dropout_prob=0.5, attention_probs_dropout_prob=0.5)   # p = 0.5: training activations are 2 x the eval ones (the 8-bit kernel wants K >= 384)
    torch.manual_seed(8)
    m = E.EcgVit(config=conf, compute_dtype=BF16, fp8_linear=True).cuda().train()
    x, y = O.synthetic_batch(12, length=5000, seed=8)
    x, y = x.cuda(), y.cuda()
    eng = m._engine()
    for _ in range(2):
        m(sample_values=x, labels=y).loss.backward()
    m(sample_values=x, labels=y)                       # its begin_step installed the scales of the second training pass
    train_scales = eng.f8_scale.clone()
    hact_site = 3                                      # layer 0: e4m3 copy of the FFN hidden activation (dropout-rescaled in training)
    m.eval()
    with torch.no_grad():
        m(sample_values=x)
        m(sample_values=x)                             # second eval pass: scales now come from an eval pass
    eval_scales = eng.f8_scale.clone()
    assert float(eval_scales[hact_site]) < 0.8 * float(train_scales[hact_site])       # what a training pass must not inherit
    m.train()
    out = m(sample_values=x, labels=y)
    after = eng.f8_scale.clone()
    fwd_sites = [8 * i + k for i in range(2) for k in range(4)]
    assert torch.equal(after[fwd_sites], eval_scales[fwd_sites]) is False
    # the scales in force are the ones the last TRAINING pass left (its own amax -> the training-size range), not the eval pass's
    assert float(after[hact_site]) > 0.8 * float(train_scales[hact_site])
    assert torch.isfinite(out.loss)
    out.loss.backward()
    m(sample_values=x, labels=y)                       # train -> train: plain delayed scaling again
    assert float(eng.f8_scale[hact_site]) > 0.8 * float(train_scales[hact_site])
