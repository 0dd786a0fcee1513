"""-m gpu: the configuration that is actually TIMED -- dropout 0.1 at all five sites per layer (reference `models/ecg_vit.py:113-114`: vit-pytorch's
`dropout` <- hidden_dropout_prob = 0.1 on attention probabilities, `to_out`, FFN hidden and FFN output; `emb_dropout` <- attention_probs_dropout_prob = 0.1)
-- held against the CPU oracle NUMBER FOR NUMBER.

Dropout masks of the HIP path are a pure function of (seed, site, element).  After a HIP forward the test exports the multipliers the kernels applied
(`hiputil.export_dropout_masks`: hidden / embedding sites through `ecgvit_dropout_apply` on ones, the fused attention kernel's probability mask observed
through the kernel itself with one-hot V), checks that the engine's own activations carry exactly those masks, INJECTS them into the oracle's five
`nn.Dropout` sites (`oracle.vit_oracle.inject_dropout`) and compares loss / outputs / every gradient:

  f32 HIP path  (exact p, 16-bit pair masks)                                        : <= 1e-4 relative (north_star tolerance)
  bf16 HIP path (quad 8-bit masks at p' = 26/256, e4m3 saved FFN tensor, 3-term erf): loss <= 2e-2, whole-gradient cosine >= 0.98, every tensor >= 0.95

at the base layer shape (d 768, 12 heads, f 3072, 12 x 5000 samples, patch 20), 2 layers, 10 records = 2 510 / 2 500 token rows (>= 2 048: the large A.B^T kernels,
the e4m3 saved tensor and the persistent attention backward are on this path), for the supervised step AND the masked pre-train step; at the LARGE layer shape
(d 1024, 16 heads, patch 10 -> 501 / 500 tokens, 16 records: two key windows in the backward, the STREAMED attention forward -- whose mask is observed through the
streamed kernel itself); and ONE full-depth base model (12 layers, 61 injected mask tensors, 140 gradient tensors, f32 <= 1e-4).
"""
import pytest
import torch

from hiputil import rel_err, max_err, export_dropout_masks, assert_engine_tensors_carry_masks
from oracle import vit_oracle as O
import ecg_representation_learning_amd as E

pytestmark = pytest.mark.gpu
F32, BF16 = torch.float32, torch.bfloat16
SHAPES = {
    'base': dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072),                     # 251 / 250 tokens
    'large': dict(hidden_size=1024, num_attention_heads=16, intermediate_size=4096, patch_size=10),    # 501 / 500 tokens: two key windows; 16 records x 16 heads = an item
}                                                                                                      # per CU: the STREAMED attention forward is on this path
P_HID, P_EMB = 0.1, 0.1   # reference defaults, models/ecg_vit.py:38-39


def _conf(name='base', **kw):
    return E.EcgVitConfig(**{**dict(max_signal_length=5000, patch_size=20, num_hidden_layers=2, hidden_dropout_prob=P_HID,
                                     attention_probs_dropout_prob=P_EMB), **SHAPES[name], **kw})


def _cos(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _check_mask_rates(masks, dtype):
    """the exported multipliers are 0 or the site's rescale, at the site's keep rate"""
    pa = round(256 * P_HID) / 256 if dtype == BF16 else P_HID
    inv = 1.0 / (1.0 - pa)
    for mk in [dict(emb=masks['emb'])] + masks['layers']:
        for k, m in mk.items():
            nz = m[m != 0]
            assert float((nz - inv).abs().max()) < 1e-6 * inv, k
            assert abs(float((m != 0).double().mean()) - (1 - pa)) < 3e-3, (k, float((m != 0).double().mean()))


@pytest.mark.parametrize('name,layers,B,dtype', [('base', 2, 10, F32), ('base', 2, 10, BF16), ('large', 2, 16, F32), ('large', 2, 16, BF16), ('base', 12, 10, F32)])
def test_supervised_step_dropout_01_vs_cpu_oracle_with_injected_masks(name, layers, B, dtype):
    """(base, 12 layers, f32): the FULL-depth model under dropout 0.1 -- 61 injected mask tensors -- every one of the 140 gradient tensors <= 1e-4"""
    torch.set_num_threads(min(32, torch.get_num_threads()))
    conf = _conf(name, num_hidden_layers=layers)
    torch.manual_seed(5)
    ref = O.OracleEcgVit(config=conf).train()
    m = E.EcgVit(config=conf, compute_dtype=dtype)
    m.load_state_dict(ref.state_dict())
    m.cuda().train()
    x, y = O.synthetic_batch(B, length=5000, seed=23)
    out = m(sample_values=x.cuda(), labels=y.cuda())
    eng = m._engine()
    assert eng.saved['ph'] == P_HID and eng.saved['pe'] == P_EMB and eng.saved['seed'] != 0 and B * eng.N >= 2048
    if dtype == BF16:
        assert eng._aux8(B * eng.N) and eng.act['layers'][0]['hpre'].dtype == torch.uint8   # the e4m3 saved tensor of the timed configuration
    masks = export_dropout_masks(eng)
    _check_mask_rates(masks, dtype)
    assert_engine_tensors_carry_masks(eng, masks)
    out.loss.backward()
    # the oracle under the SAME dropout realisation
    O.inject_dropout(ref.vit, masks)
    o_ref = ref(sample_values=x, labels=y)
    o_ref.loss.backward()
    # ... which is not the p = 0 result: the comparison below would fail without the injection
    O.inject_dropout(ref.vit, None)
    ref.eval()
    with torch.no_grad():
        o_nodrop = ref(sample_values=x, labels=y)
    assert max_err(o_nodrop.logits, o_ref.logits) > 1e-2
    lerr = abs(float(out.loss.detach()) - float(o_ref.loss.detach())) / float(o_ref.loss.detach())
    pr = dict(ref.named_parameters())
    if dtype == F32:
        worst = max(rel_err(p.grad, pr[k].grad) for k, p in m.named_parameters())
        print(f'[{name} x {layers} f32, dropout 0.1] loss rel {lerr:.2e}, logits max {max_err(out.logits, o_ref.logits):.2e}, worst gradient tensor rel {worst:.2e}')
        assert lerr < 1e-4, lerr
        assert max_err(out.logits, o_ref.logits) < 1e-4
        for k, p in m.named_parameters():
            assert rel_err(p.grad, pr[k].grad) < 1e-4, (k, rel_err(p.grad, pr[k].grad))
    else:
        g16 = torch.cat([p.grad.flatten() for _, p in m.named_parameters()])
        gref = torch.cat([pr[k].grad.flatten() for k, _ in m.named_parameters()])
        worst = min(_cos(p.grad, pr[k].grad) for k, p in m.named_parameters())
        print(f'[{name} x {layers} bf16, dropout 0.1] loss rel {lerr:.2e}, logits max {max_err(out.logits, o_ref.logits):.2e}, gradient cosine {_cos(g16, gref):.5f}, worst tensor {worst:.5f}')
        # (observed on MI355X: loss 1.5e-5 / 2.2e-5, logits 6e-3 / 8e-3, whole-gradient cosine 0.99999, worst tensor 0.99993 -- the bounds asked for are 2e-2,
        # 0.98 and 0.95; held ten times tighter here so that a mask or saved-tensor mismatch of a few elements in a thousand fails)
        assert lerr < 2e-3, lerr
        assert max_err(out.logits, o_ref.logits) < 0.05
        assert _cos(g16, gref) > 0.999, _cos(g16, gref)
        for k, p in m.named_parameters():
            assert _cos(p.grad, pr[k].grad) > 0.99, (k, _cos(p.grad, pr[k].grad))


@pytest.mark.parametrize('name,B,dtype', [('base', 10, F32), ('base', 10, BF16), ('large', 16, BF16)])
def test_masked_step_dropout_01_vs_cpu_oracle_with_injected_masks(name, B, dtype):
    """the masked pre-train step (250 tokens, no CLS row; embedding dropout on the [B, n, d] token slab) through the FUSED train step
    (`HipTrainStep.step_masked`: the launches `bench.py --objective masked` times), its flat gradient buffer against the oracle's gradients"""
    torch.set_num_threads(min(32, torch.get_num_threads()))
    conf = _conf(name)
    n = 5000 // conf.patch_size
    torch.manual_seed(6)
    ref = O.OracleMaskedEcgVit(O.OracleEcgVit(config=conf)).train()
    mm = E.MaskedEcgVit(E.EcgVit(config=conf, compute_dtype=dtype), mask_ratio=0.5)
    mm.load_state_dict(ref.state_dict(), strict=True)
    mm.cuda().train()
    x, _ = O.synthetic_batch(B, length=5000, seed=29)
    idx = mm.random_mask_indices(B, generator=torch.Generator().manual_seed(4))
    enc = mm.encoder
    step = E.HipTrainStep(mm, dict(n_step=10, learning_rate=0.0, weight_decay=0.0))   # lr 0: the step leaves the weights where the oracle has them
    loss, pred = step.step_masked(x.cuda(), idx)
    step.finish()
    eng = enc._engine()
    assert eng.T == n and eng.saved['masked'] and eng.saved['ph'] == P_HID and eng.saved['pe'] == P_EMB and eng.saved['seed'] != 0
    masks = export_dropout_masks(eng)
    _check_mask_rates(masks, dtype)
    assert_engine_tensors_carry_masks(eng, masks)
    O.inject_dropout(ref.encoder.vit, masks)
    o_ref = ref(x, idx)
    o_ref.loss.backward()
    lerr = abs(float(loss) - float(o_ref.loss.detach())) / float(o_ref.loss.detach())
    pr = dict(ref.named_parameters())
    names = {'mask_token': 'pretrain.mask_token', 'to_pixels.weight': 'pretrain.to_pixels.weight', 'to_pixels.bias': 'pretrain.to_pixels.bias'}
    got, want = [], []
    for k, q in pr.items():
        flat_name = names.get(k, k[len('encoder.'):] if k.startswith('encoder.') else k)
        g = enc._layout.view(enc._gflat, flat_name)
        if q.grad is None:   # cls_token / classification head take no part in this objective
            assert float(g.abs().max()) == 0.0, k
            continue
        got.append(g.flatten())
        want.append(q.grad.flatten())
        if dtype == F32:
            assert rel_err(g, q.grad) < 1e-4, (k, rel_err(g, q.grad))
        else:
            assert _cos(g, q.grad) > 0.95, (k, _cos(g, q.grad))
    predv = pred.float().view(B, n // 2, -1)
    if dtype == F32:
        assert lerr < 1e-4, lerr
        assert rel_err(predv, o_ref.logits) < 1e-4
    else:
        assert lerr < 2e-2, lerr
        assert rel_err(predv, o_ref.logits) < 3e-2
        assert _cos(torch.cat(got), torch.cat(want)) > 0.98
