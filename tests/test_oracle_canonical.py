"""CPU: the oracle's transformer trunk (oracle/vit_oracle.py: a restatement of vit-pytorch 0.33.2's PreNorm / Attention / FeedForward, which is not installable
here: "parity unpinned") against an INDEPENDENT implementation of the same published block -- torch.nn.TransformerEncoderLayer(norm_first=True,
activation='gelu'), written by other people from the same definition (pre-LN residual block, bias-free packed QKV with rows [q; k; v] head-major, softmax(QK^T /
sqrt(dh)) V, erf GELU MLP).  It does NOT pin the oracle to vit-pytorch's code; it does show that the restated arithmetic is the canonical pre-LN encoder block
and not a private variant: same weights in, same activations and same gradients out, in float64."""
import os
import sys

import torch

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_oracle_trunk_is_the_canonical_pre_ln_encoder():
    from oracle import vit_oracle as O
    torch.manual_seed(3)
    dim, depth, heads, dh, mlp = 64, 3, 4, 16, 160
    trunk = O._Transformer(dim, depth, heads, dh, mlp, 0.0).double()
    layer = torch.nn.TransformerEncoderLayer(d_model=dim, nhead=heads, dim_feedforward=mlp, dropout=0.0, activation='gelu', batch_first=True, norm_first=True)
    enc = torch.nn.TransformerEncoder(layer, num_layers=depth, enable_nested_tensor=False).double()
    with torch.no_grad():
        for (attn, ff), lay in zip(trunk.layers, enc.layers):
            for p in list(attn.parameters()) + list(ff.parameters()):
                p.copy_(torch.randn_like(p) * 0.2)
            lay.self_attn.in_proj_weight.copy_(attn.fn.to_qkv.weight)      # rows [q; k; v], each head-major: the same packing
            lay.self_attn.in_proj_bias.zero_()                             # vit-pytorch's to_qkv has no bias
            lay.self_attn.out_proj.weight.copy_(attn.fn.to_out[0].weight)
            lay.self_attn.out_proj.bias.copy_(attn.fn.to_out[0].bias)
            lay.norm1.weight.copy_(attn.norm.weight); lay.norm1.bias.copy_(attn.norm.bias)
            lay.norm2.weight.copy_(ff.norm.weight); lay.norm2.bias.copy_(ff.norm.bias)
            lay.linear1.weight.copy_(ff.fn.net[0].weight); lay.linear1.bias.copy_(ff.fn.net[0].bias)
            lay.linear2.weight.copy_(ff.fn.net[3].weight); lay.linear2.bias.copy_(ff.fn.net[3].bias)
    x = torch.randn(3, 37, dim, dtype=torch.float64)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    trunk.train(); enc.train()          # (train mode keeps torch off its inference fast path; dropout is 0)
    ya, yb = trunk(xa), enc(xb)
    assert float((ya - yb).abs().max()) < 1e-11, float((ya - yb).abs().max())
    g = torch.randn_like(ya)
    ya.backward(g); yb.backward(g)
    assert float((xa.grad - xb.grad).abs().max()) < 1e-11
    for (attn, ff), lay in zip(trunk.layers, enc.layers):
        assert float((attn.fn.to_qkv.weight.grad - lay.self_attn.in_proj_weight.grad).abs().max()) < 1e-10
        assert float((ff.fn.net[0].weight.grad - lay.linear1.weight.grad).abs().max()) < 1e-10
        assert float((ff.norm.weight.grad - lay.norm2.weight.grad).abs().max()) < 1e-10
    assert os.path.exists(os.path.join(ROOT, 'oracle', 'vit_oracle.py'))


def test_injected_dropout_replays_torchs_own_realisation():
    """`InjectedDropout` (the oracle's five dropout sites; reference models/ecg_vit.py:113-114 sets their p): un-injected it IS nn.Dropout (same
    generator stream); with the multipliers torch itself drew at every site recorded and injected back, a second pass reproduces the first to
    rounding (1e-6) -- loss, logits and every gradient -- in train mode, and also in eval mode (injection overrides the mode).  So the injection points are all
    the places where the model draws a mask, in the shapes `inject_dropout` documents."""
    from conftest import Cfg
    from oracle import vit_oracle as O
    conf = Cfg(max_signal_length=400, patch_size=20, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
               hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.2)
    torch.manual_seed(11)
    ref = O.OracleEcgVit(config=conf).train()
    x, y = O.synthetic_batch(3, length=400, seed=5)
    sites = O.dropout_sites(ref.vit)
    flat = [('emb', sites['emb'])] + [(f'{i}.{k}', m) for i, s in enumerate(sites['layers']) for k, m in s.items()]
    assert len(flat) == 1 + 4 * 2 and all(isinstance(m, O.InjectedDropout) for _, m in flat)
    assert sum(isinstance(m, torch.nn.Dropout) for m in ref.modules()) == len(flat)          # no dropout module outside the five kinds of site
    rec = {}
    hooks = [m.register_forward_hook(lambda mod, inp, out, k=k: rec.__setitem__(k, torch.where(inp[0] != 0, out / inp[0], torch.ones_like(out)).detach()))
             for k, m in flat]
    torch.manual_seed(99)
    o1 = ref(sample_values=x, labels=y)
    o1.loss.backward()
    g1 = [p.grad.clone() for p in ref.parameters()]
    for h in hooks:
        h.remove()
    # un-injected == nn.Dropout on the same generator state
    torch.manual_seed(99)
    o1b = ref(sample_values=x, labels=y)
    assert torch.equal(o1.logits, o1b.logits)
    p_emb = float((rec['emb'] == 0).float().mean())
    assert 0.1 < p_emb < 0.3 and abs(float(rec['0.ffn'].max()) - 1 / 0.9) < 1e-5
    masks = dict(emb=rec['emb'], layers=[{k: rec[f'{i}.{k}'] for k in ('probs', 'out', 'ffn', 'down')} for i in range(2)])
    assert masks['layers'][0]['probs'].shape == (3, 2, 21, 21) and masks['emb'].shape == (3, 21, 32) and masks['layers'][1]['ffn'].shape == (3, 21, 64)
    for mode in ('train', 'eval'):
        getattr(ref, mode)()
        O.inject_dropout(ref.vit, masks)
        ref.zero_grad()
        o2 = ref(sample_values=x, labels=y)
        o2.loss.backward()
        assert torch.allclose(o2.logits, o1.logits, rtol=0, atol=1e-6) and abs(float(o2.loss) - float(o1.loss)) < 1e-7, mode   # (the recorded ratio out / in is the multiplier up to one rounding)
        for a, b in zip(g1, [p.grad for p in ref.parameters()]):
            assert torch.allclose(a, b, rtol=0, atol=1e-7), mode
    O.inject_dropout(ref.vit, None)
    with torch.no_grad():
        assert not torch.equal(ref(sample_values=x).logits, o1.logits)                       # injection removed: eval mode is dropout-free again


def test_oracle_trunk_matches_huggingface_vit_layers():
    """a SECOND independent implementation of the same published block: HuggingFace `transformers` `ViTLayer` (pre-LN: layernorm_before -> self-attention ->
    residual -> layernorm_after -> GELU MLP -> residual; separate bias-free q / k / v projections = the three row blocks of vit-pytorch's packed `to_qkv`).
    Same weights in, same activations and input gradients out -- to 1e-6 in float64 (HF's eager attention takes its softmax in float32).  Like the
    nn.TransformerEncoderLayer check above this does NOT pin the oracle to vit-pytorch's code (the package is not installable here: parity unpinned); it shows
    once more that the restated arithmetic is the canonical pre-LN ViT block."""
    import pytest
    from oracle import vit_oracle as O
    try:
        from transformers import ViTConfig
        from transformers.models.vit.modeling_vit import ViTLayer
    except Exception as e:   # pragma: no cover
        pytest.skip(f'transformers ViT not importable: {e}')
    torch.manual_seed(3)
    dim, depth, heads, dh, mlp = 64, 3, 4, 16, 160
    cfg = ViTConfig(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads, intermediate_size=mlp, hidden_act='gelu', hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0, layer_norm_eps=1e-5, qkv_bias=False)
    cfg._attn_implementation = 'eager'
    layers = torch.nn.ModuleList([ViTLayer(cfg) for _ in range(depth)]).double()
    if not hasattr(layers[0].attention, 'q_proj') or not hasattr(layers[0], 'mlp'):
        pytest.skip('this transformers release lays ViTLayer out differently (written against 5.15: attention.{q,k,v,o}_proj, mlp.fc1 / fc2)')
    trunk = O._Transformer(dim, depth, heads, dh, mlp, 0.0).double()
    with torch.no_grad():
        for (attn, ff), lay in zip(trunk.layers, layers):
            for p in list(attn.parameters()) + list(ff.parameters()):
                p.copy_(torch.randn_like(p) * 0.2)
            W, sa = attn.fn.to_qkv.weight, lay.attention
            sa.q_proj.weight.copy_(W[:dim]); sa.k_proj.weight.copy_(W[dim:2 * dim]); sa.v_proj.weight.copy_(W[2 * dim:])
            sa.o_proj.weight.copy_(attn.fn.to_out[0].weight); sa.o_proj.bias.copy_(attn.fn.to_out[0].bias)
            lay.layernorm_before.weight.copy_(attn.norm.weight); lay.layernorm_before.bias.copy_(attn.norm.bias)
            lay.layernorm_after.weight.copy_(ff.norm.weight); lay.layernorm_after.bias.copy_(ff.norm.bias)
            lay.mlp.fc1.weight.copy_(ff.fn.net[0].weight); lay.mlp.fc1.bias.copy_(ff.fn.net[0].bias)
            lay.mlp.fc2.weight.copy_(ff.fn.net[3].weight); lay.mlp.fc2.bias.copy_(ff.fn.net[3].bias)
    x = torch.randn(3, 37, dim, dtype=torch.float64)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    trunk.train(); layers.train()
    ya, h = trunk(xa), xb
    for lay in layers:
        out = lay(h)
        h = out[0] if isinstance(out, tuple) else out
    assert float((ya - h).detach().abs().max()) < 1e-6
    g = torch.randn_like(ya)
    ya.backward(g); h.backward(g)
    assert float((xa.grad - xb.grad).abs().max()) < 1e-6
    for (attn, ff), lay in zip(trunk.layers, layers):
        assert float((attn.fn.to_qkv.weight.grad[:dim] - lay.attention.q_proj.weight.grad).abs().max()) < 1e-5
        assert float((ff.fn.net[0].weight.grad - lay.mlp.fc1.weight.grad).abs().max()) < 1e-5
