"""helpers for the -m gpu tests: call the C-ABI directly with torch device tensors"""
import torch

import ecg_representation_learning_amd as E
from ecg_representation_learning_amd import hip
from ecg_representation_learning_amd.hip import lib, check, ptr, stream


def tools_lib():
    """the TOOLS build of the library (tools/ecgvit_hip_tools.h): test-only entry points -- the fragment-layout probe and the one-item attention
    backward the persistent kernel is held against.  Built by __graft_entry__.build(); never loaded by the product package."""
    import os
    import sys
    tools_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools')
    if tools_dir not in sys.path:
        sys.path.insert(0, tools_dir)
    import toolslib
    return toolslib.tools_lib()


_keepalive = []  # device copies made by dev() stay alive until the test ends (raw pointers are handed to the C-ABI)


def dev(t, dtype=None):
    t = torch.as_tensor(t)
    if dtype is not None:
        t = t.to(dtype)
    d = t.contiguous().cuda()
    _keepalive.append(d)
    return d


def release():
    _keepalive.clear()


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max())


def gelu(x):
    return torch.nn.functional.gelu(x)


def gelu_grad(x):
    x = x.double()
    cdf = 0.5 * (1 + torch.erf(x / 2 ** 0.5))
    pdf = torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    return (cdf + x * pdf)


# ---------------------------------------------------------------------------------------------------------------------
# dropout realisation of a step the engine just ran, exported for injection into the CPU oracle (oracle.vit_oracle.inject_dropout)
# ---------------------------------------------------------------------------------------------------------------------
def _site_mult(count, p, seed, dtype):
    """multipliers of a hidden / embedding site over `count` contiguous elements: `ecgvit_dropout_apply` on ones (the GEMM epilogues draw the
    same mask: test_large_gemm_dropout_mask_equals_dropout_apply; the engine's own tensors are checked against it by the caller).  bf16 sites
    apply p' = round(256 p) / 256 with the EXACT f32 rescale 256 / (256 - round(256 p)) before rounding the product, so the bf16-rounded 'one
    times multiplier' read back here is replaced by that constant"""
    ones = torch.ones(count, device='cuda', dtype=dtype)
    out = torch.empty_like(ones)
    check(lib().ecgvit_dropout_apply(ptr(ones), ptr(out), count, float(p), int(seed), hip.code(dtype), stream()), 'dropout_apply')
    out = out.float()
    if dtype == torch.bfloat16:
        t = round(256 * p)
        out = (out != 0).float() * (256.0 / (256.0 - t))
    return out.cpu()


def _attn_prob_mult_bf16(B, h, T, p, seed):
    """multipliers the FUSED bf16 attention kernel applies to its probabilities, observed through the kernel itself: Q = K = 0 makes every
    probability 1/T, V = one-hot of (key - 64 w) for the keys of window w exposes P~[q, 64 w + j] as output column j.  (B, h, T, T) f32"""
    dh, d = 64, h * 64
    t = round(256 * p)
    inv = 256.0 / (256.0 - t)
    mult = torch.zeros(B, h, T, T)
    out = torch.empty(B * T, d, device='cuda', dtype=torch.bfloat16)
    lse = torch.empty(B * h * T, device='cuda')
    for w in range((T + 63) // 64):
        qkv = torch.zeros(B, T, 3, h, dh)
        k = torch.arange(64 * w, min(T, 64 * w + 64))
        qkv[:, k, 2, :, k - 64 * w] = 1.0
        qd = qkv.reshape(B * T, 3 * d).to(torch.bfloat16).cuda()
        check(lib().ecgvit_attention_fwd(ptr(qd), ptr(out), ptr(lse), B, T, h, dh, dh ** -0.5, float(p), int(seed), hip.BF16, stream()), 'attention_fwd')
        o = out.float().cpu().view(B, T, h, dh).permute(0, 2, 1, 3) * T          # [b, head, q, j] = multiplier of key 64 w + j (bf16-rounded)
        o = o[..., :len(k)]
        assert bool(((o == 0) | ((o - inv).abs() < 2e-2 * inv)).all()), 'the fused kernel returned something other than 0 or P / keep-rate'
        mult[..., 64 * w:64 * w + len(k)] = (o != 0).float() * inv
    return mult


def export_dropout_masks(eng):
    """{'emb', 'layers': [{'probs', 'out', 'ffn', 'down'}]} multipliers (CPU f32) of the forward `eng` (a VitEngine) ran last -- seeds and element
    order as engine.py issues them: embedding seed + 1 over x0 [B*T, d]; layer i (s0 = seed + 100 (i + 1)): probabilities s0 + 1 over
    [B, h, T, T], to_out s0 + 2 over [B*T, d], FFN hidden s0 + 3 over [B*T, f], FFN output s0 + 4 over [B*T, d]"""
    sv = eng.saved
    B, ph, pe, seed, T = sv['B'], sv['ph'], sv['pe'], sv['seed'], eng.T
    d, f, h = eng.d, eng.f, eng.h
    dt = eng.dtype
    M = B * T
    masks = dict(emb=_site_mult(M * d, pe, seed + 1, dt).view(B, T, d) if pe > 0 else None, layers=[])
    for i in range(eng.Ly):
        s0 = seed + 100 * (i + 1)
        if ph <= 0:
            masks['layers'].append(dict())
            continue
        if dt == torch.float32:
            probs = _site_mult(B * h * T * T, ph, s0 + 1, dt).view(B, h, T, T)
        else:
            probs = _attn_prob_mult_bf16(B, h, T, ph, s0 + 1)
        masks['layers'].append(dict(probs=probs, out=_site_mult(M * d, ph, s0 + 2, dt).view(B, T, d), ffn=_site_mult(M * f, ph, s0 + 3, dt).view(B, T, f),
                                    down=_site_mult(M * d, ph, s0 + 4, dt).view(B, T, d)))
    return masks


def assert_engine_tensors_carry_masks(eng, masks):
    """the activations the forward left behind show the exported masks: x0 / hact are zero exactly where embedding / FFN-hidden units were dropped,
    x1 / x2 equal their residual input exactly where the to_out / FFN-output units were dropped (the converse holds up to coincidences: a kept
    value that is itself zero or rounds away against the residual)"""
    a = eng.act
    B, T = eng.saved['B'], eng.T

    def dropped_implies(flag, m, what, min_frac):
        drop = (m.reshape(flag.shape) == 0)
        assert bool(flag[drop].all()), what
        assert float((flag & ~drop).float().mean()) < min_frac, (what, float((flag & ~drop).float().mean()))
    if masks['emb'] is not None:
        dropped_implies((a['x0'].float() == 0).cpu(), masks['emb'], 'embedding dropout', 1e-3)
    X = a['x0']
    for L, mk in zip(a['layers'], masks['layers']):
        if mk:
            dropped_implies((L['x1'] == X).cpu(), mk['out'], 'to_out dropout', 0.12)   # (bf16: a kept value below half an ulp of the residual rounds away: 5.5 % at d = 1024)
            dropped_implies((L['hact'].float() == 0).cpu(), mk['ffn'], 'FFN hidden dropout', 1e-3)
            dropped_implies((L['x2'] == L['x1']).cpu(), mk['down'], 'FFN output dropout', 0.12)
        X = L['x2']
