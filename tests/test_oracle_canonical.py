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
