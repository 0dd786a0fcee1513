#!/usr/bin/env python3
"""race / exactness stress of the streaming kernels (counted vmcnt schedules): random shapes, small-integer operands (exact f32 sums),
repeated launches, persistent attention backward vs the one-item kernel.  python tools/stress.py [seconds]"""
import os, sys, time, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_representation_learning_amd import hip
from ecg_representation_learning_amd.hip import lib, check, ptr, stream
from toolslib import tools_lib
import bench as _bench  # noqa: E402
print('kernel_source_sha16:', _bench.kernel_source_hash(), '(sources of the library build measured: tools/check_profiles.py holds committed tables to the round\'s bench line)', flush=True)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
only = sys.argv[2] if len(sys.argv) > 2 else None   # e.g. `python tools/stress.py 300 attnfwd`: one kind of case only
rng = random.Random(1234)
bf = torch.bfloat16
t_end = time.time() + budget
n_nt = n_tn = n_at = n_aux8 = bad = 0
while time.time() < t_end:
    kind = only or rng.choice(['nt', 'nt', 'ntlin', 'ntaux8', 'nt8', 'nt8emit', 'tn', 'tn', 'tn8', 'attn', 'attn8', 'attnfwd'])
    if kind == 'nt':
        M = rng.choice([2048, 4133, 20000, 66000, 128512]) + rng.randrange(0, 256)
        N = rng.choice([128, 240, 256, 520, 768, 776, 2304, 3072])
        K = 64 * rng.randrange(3, 50)
        A = torch.randint(-3, 4, (M, K), device='cuda').to(bf); B = torch.randint(-3, 4, (N, K), device='cuda').to(bf)
        ref = A.float() @ B.float().t()
        out_f32 = rng.random() < 0.3
        C = torch.empty(M, N, device='cuda', dtype=torch.float32 if out_f32 else bf)
        want = ref if out_f32 else ref.to(bf)
        for _ in range(4):
            C.fill_(float('nan'))
            hip.gemm(hip.GEMM_NT, A, B, C, M, N, K, K, K, N, tiles_per_workgroup=rng.choice([0, 0, 1, 2, 3]))
            if not torch.equal(C, want):
                bad += 1; print('NT MISMATCH', M, N, K, out_f32, int((C != want).sum()), flush=True)
        n_nt += 1
    elif kind == 'ntlin':    # bias + residual on exact operands: the four-wave body (K >= 768), persistent and chunked, against the reference
        M = rng.choice([2048, 4133, 20000, 66000]) + rng.randrange(0, 256)
        N = rng.choice([256, 520, 768, 776, 2304])
        K = 64 * rng.randrange(12, 50)
        A = torch.randint(-3, 4, (M, K), device='cuda').to(bf); B = torch.randint(-3, 4, (N, K), device='cuda').to(bf)
        bias = torch.randint(-4, 5, (N,), device='cuda').float(); res = torch.randint(-8, 9, (M, N), device='cuda').to(bf)
        want = (A.float() @ B.float().t() + bias + res.float()).to(bf)
        C = torch.empty(M, N, device='cuda', dtype=bf)
        for tpw in (0, 0, 0, 2):
            C.fill_(float('nan'))
            hip.gemm(hip.GEMM_NT, A, B, C, M, N, K, K, K, N, epilogue=hip.EPI_BIAS | hip.EPI_RESIDUAL, bias=bias, residual=res, ldr=N, tiles_per_workgroup=tpw)
            if not torch.equal(C, want):
                bad += 1; print('NT-LIN MISMATCH', M, N, K, tpw, int((C != want).sum()), flush=True)
        n_nt += 1
    elif kind == 'ntaux8':   # the x-aux input-gradient body with the e4m3 saved tensor (its rows are requested a main loop ahead, across tile boundaries):
        # small-integer operands and exactly representable multipliers -- outputs and column sums are exact whatever the order
        M = rng.choice([2048, 4133, 20000, 66000]) + rng.randrange(0, 256)
        N = rng.choice([256, 520, 768, 2048, 3072])
        K = 64 * rng.randrange(3, 7)   # (the flag needs K >= 192)
        A = torch.randint(-1, 2, (M, K), device='cuda').to(bf); B = torch.randint(-1, 2, (N, K), device='cuda').to(bf)
        auxv = torch.tensor([0.0, 0.5, 1.0, 1.0, 2.0], device='cuda')[torch.randint(0, 5, (M, N), device='cuda')]
        aux8 = auxv.to(torch.float8_e4m3fn).view(torch.uint8)
        want = ((A.float() @ B.float().t()) * auxv).to(bf)
        wcs = want.float().sum(0)
        ws = torch.empty(max(lib().ecgvit_colsum_workspace(M, N), 8 * ((M + 255) // 256) * N), dtype=torch.uint8, device='cuda')
        C = torch.empty(M, N, device='cuda', dtype=bf); cs = torch.empty(N, device='cuda')
        for tpw in (0, 0, 2):
            C.fill_(float('nan')); cs.fill_(float('nan'))
            try:
                hip.gemm(hip.GEMM_NT, A, B, C, M, N, K, K, K, N, epilogue=hip.EPI_MUL_AUX | hip.EPI_COLSUM | hip.EPI_AUX8, aux=aux8, ldaux=N, workspace=ws, colsum_out=cs,
                         tiles_per_workgroup=tpw)
            except RuntimeError:   # (shapes the streaming kernel does not take reject the flag: not a failure)
                break
            n_aux8 += 1
            if not (torch.equal(C, want) and torch.equal(cs, wcs)):
                bad += 1; print('NT-AUX8 MISMATCH', M, N, K, tpw, int((C != want).sum()), int((cs != wcs).sum()), flush=True)
        n_nt += 1
    elif kind == 'nt8':      # 8-bit operands (e4m3 x e4m3 / e5m2 x e4m3), small integers: exact
        M = rng.choice([2048, 4133, 20000, 66000, 128256]) + rng.randrange(0, 256)
        N = rng.choice([128, 256, 520, 768, 1024, 3072, 4096])
        K = 128 * rng.randrange(3, 33)
        afmt, adt = rng.choice([(hip.FP8_E4M3, torch.float8_e4m3fn), (hip.BF8_E5M2, torch.float8_e5m2)])
        A = torch.randint(-2, 3, (M, K), device='cuda').float().to(adt); B = torch.randint(-2, 3, (N, K), device='cuda').float().to(torch.float8_e4m3fn)
        want = (A.float() @ B.float().t()).to(bf)
        C = torch.empty(M, N, device='cuda', dtype=bf)
        one = torch.ones(1, device='cuda')
        for _ in range(4):
            C.fill_(float('nan'))
            hip.gemm(hip.GEMM_NT, A.view(torch.uint8), B.view(torch.uint8), C, M, N, K, K, K, N, fp8_format=afmt, scale_a=one, scale_b=one,
                     tiles_per_workgroup=rng.choice([0, 0, 2]))
            if not torch.equal(C, want):
                bad += 1; print('NT8 MISMATCH', M, N, K, afmt, int((C != want).sum()), flush=True)
        n_nt += 1
    elif kind == 'nt8emit':  # the FFN-wide emitting epilogues of the 8-bit kernel on random shapes: writing form vs no-output form (ABI 5), bit for bit
        M = rng.choice([2048, 4133, 20000, 66000]) + rng.randrange(0, 256)
        N = rng.choice([256, 520, 1024, 3072, 4096])
        K = 128 * rng.randrange(3, 17)
        up = rng.random() < 0.5
        afmt, adt = (hip.FP8_E4M3, torch.float8_e4m3fn) if up else (hip.BF8_E5M2, torch.float8_e5m2)
        A = (torch.randn(M, K, device='cuda') * 2).to(adt); B = (torch.randn(N, K, device='cuda') * 0.2).to(torch.float8_e4m3fn)
        s1 = torch.tensor([0.5], device='cuda'); qs = torch.tensor([0.02 if up else 1e-3], device='cuda')
        bias = torch.randn(N, device='cuda'); auxin = (torch.rand(M, N, device='cuda') * 1.2).to(bf)
        ws = torch.empty(max(lib().ecgvit_colsum_workspace(M, N), 8 * ((M + 255) // 256) * N), dtype=torch.uint8, device='cuda')
        drop = up and rng.random() < 0.5
        res = []
        for no_out in (False, True, True):
            C = torch.full((M, N), 7.0, device='cuda', dtype=bf); q8 = torch.full((M, N), 0x7F, dtype=torch.uint8, device='cuda')
            am, cs = torch.zeros(1, device='cuda'), torch.zeros(N, device='cuda')
            aux = torch.zeros(M, N, device='cuda', dtype=bf) if up else auxin
            epi = (hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_GELU_GRAD_AUX | (hip.EPI_DROPOUT if drop else 0)) if up else (hip.EPI_MUL_AUX | hip.EPI_COLSUM)
            hip.gemm(hip.GEMM_NT, A.view(torch.uint8), B.view(torch.uint8), None if no_out else C, M, N, K, K, K, N, fp8_format=afmt, scale_a=s1, scale_b=s1,
                     epilogue=epi | hip.EPI_QUANT_OUT | (hip.EPI_NO_OUT if no_out else 0), bias=bias if up else None, aux=aux, ldaux=N, dropout_p=0.1 if drop else 0.0, seed=5,
                     workspace=ws, colsum_out=None if up else cs, q8_out=q8, ldq8=N, q8_scale=qs, q8_amax=am, q8_format=hip.FP8_E4M3 if up else hip.BF8_E5M2,
                     tiles_per_workgroup=rng.choice([0, 0, 2]))
            res.append((q8, am, cs, aux.clone() if up else None, C))
        ok = all(torch.equal(res[0][i], r[i]) for r in res[1:] for i in (0, 1, 2)) and (not up or all(torch.equal(res[0][3], r[3]) for r in res[1:]))
        ok = ok and bool((res[1][4] == 7.0).all()) and float(res[0][1]) == float(res[0][4].float().abs().max())
        if not ok:
            bad += 1; print('NT8-EMIT MISMATCH', M, N, K, up, drop, flush=True)
        n_nt += 1
    elif kind == 'tn':
        M = 256 * rng.randrange(1, 13); N = 256 * rng.randrange(1, 13)
        K = rng.randrange(4096, 140000)
        A = torch.randint(-2, 3, (K, M), device='cuda').to(bf); B = torch.randint(-2, 3, (K, N), device='cuda').to(bf)
        ref = A.float().t() @ B.float()
        ws = torch.empty(max(16, hip.gemm_workspace_bytes(hip.GEMM_TN, bf, M, N, K)), dtype=torch.uint8, device='cuda')
        C = torch.empty(M, N, device='cuda')
        for _ in range(3):
            C.fill_(float('nan'))
            hip.gemm(hip.GEMM_TN, A, B, C, M, N, K, M, N, N, workspace=ws)
            if not torch.equal(C, ref):
                bad += 1; print('TN MISMATCH', M, N, K, int((C != ref).sum()), flush=True)
        n_tn += 1
    elif kind == 'tn8':     # 8-bit weight gradients (e5m2 / e4m3 x e4m3, transposed 8-bit LDS reads), small integers: exact; ragged K
        M = 256 * rng.randrange(1, 13); N = 256 * rng.randrange(1, 17)
        K = rng.randrange(4096, 140000)
        afmt, adt = rng.choice([(hip.FP8_E4M3, torch.float8_e4m3fn), (hip.BF8_E5M2, torch.float8_e5m2)])
        A = torch.randint(-2, 3, (K, M), device='cuda').float().to(adt); B = torch.randint(-2, 3, (K, N), device='cuda').float().to(torch.float8_e4m3fn)
        ref = A.float().t() @ B.float()
        ws = torch.empty(max(16, hip.gemm_workspace_bytes(hip.GEMM_TN, bf, M, N, K)), dtype=torch.uint8, device='cuda')
        C = torch.empty(M, N, device='cuda')
        one = torch.ones(1, device='cuda')
        for _ in range(3):
            C.fill_(float('nan'))
            hip.gemm(hip.GEMM_TN, A.view(torch.uint8), B.view(torch.uint8), C, M, N, K, M, N, N, workspace=ws, fp8_format=afmt, scale_a=one, scale_b=one)
            if not torch.equal(C, ref):
                bad += 1; print('TN8 MISMATCH', M, N, K, afmt, int((C != ref).sum()), flush=True)
        n_tn += 1
    elif kind == 'attnfwd':   # the streamed forward (round 6: all three workgroup forms) against the one-item kernel, random shapes: bit-identical, repeatable
        h = rng.choice([1, 2, 3, 5, 12]); N = rng.randrange(1, 513); B = rng.choice([3, 40, 90, 300, 700]) if h < 5 else rng.choice([8, 30, 64])
        p = rng.choice([0.0, 0.1, 0.3]); d = h * 64
        qkv = (torch.randn(B * N, 3 * d, device='cuda') * 1.2).to(bf)
        tl = tools_lib()
        res = []
        for v in (0, 1, 1, 2):
            tl.ecgvit_tools_attn_fwd_variant(v)
            out = torch.full((B * N, d), float('nan'), device='cuda', dtype=bf); lse = torch.full((B * h * N,), float('nan'), device='cuda')
            check(tl.ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, 64, 0.125, p, 91, hip.BF16, stream()), 'f')
            torch.cuda.synchronize(); res.append((out, lse))
        sc = torch.full((1,), 0.004, device='cuda'); res8 = []
        for v in (0, 1, 2):          # the 8-bit emitting entry point of each form, the amax slot zeroed as inside a train step
            tl.ecgvit_tools_attn_fwd_variant(v)
            out = torch.full((B * N, d), float('nan'), device='cuda', dtype=bf); lse = torch.full((B * h * N,), float('nan'), device='cuda')
            o8 = torch.full((B * N, d), 0x7F, device='cuda', dtype=torch.uint8); am = torch.zeros(1, device='cuda')
            check(tl.ecgvit_attention_fwd_q8(ptr(qkv), ptr(out), ptr(lse), B, N, h, 64, 0.125, p, 91, ptr(o8), ptr(sc), ptr(am), stream()), 'f8')
            torch.cuda.synchronize(); res8.append((out, lse, o8, float(am)))
        tl.ecgvit_tools_attn_fwd_variant(-1)
        ok8 = all(torch.equal(res[0][0].view(torch.int16), r[0].view(torch.int16)) and torch.equal(res[0][1], r[1]) and torch.equal(res8[0][2], r[2]) and r[3] == res8[0][3] for r in res8)
        if not all(torch.equal(res[0][0].view(torch.int16), r[0].view(torch.int16)) and torch.equal(res[0][1], r[1]) for r in res[1:]) or not torch.isfinite(res[0][1]).all() or not ok8:
            bad += 1; print('ATTN FWD streamed != one-item', B, h, N, p, ok8, flush=True)
        n_at += 1
    elif kind == 'attn8':   # the emitting attention kernels against the plain ones: same bf16 results bit for bit, 8-bit copies = casts of them
        h = rng.choice([1, 2, 5, 12]); N = rng.randrange(129, 513); B = rng.choice([3, 40, 90]) if h < 12 else rng.choice([8, 30])
        p = rng.choice([0.0, 0.1]); d = h * 64
        qkv = (torch.randn(B * N, 3 * d, device='cuda') * 1.2).to(bf); do = torch.randn(B * N, d, device='cuda').to(bf)
        out, out2 = (torch.empty(B * N, d, device='cuda', dtype=bf) for _ in range(2)); lse, lse2 = (torch.zeros(B * h * N, device='cuda') for _ in range(2))
        check(lib().ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, 64, 0.125, p, 77, hip.BF16, stream()), 'f')
        sc = (out.float().abs().max() / 448.0).reshape(1).clamp_min(1e-6); am = torch.zeros(1, device='cuda')
        o8 = torch.zeros(B * N, d, dtype=torch.uint8, device='cuda')
        check(lib().ecgvit_attention_fwd_q8(ptr(qkv), ptr(out2), ptr(lse2), B, N, h, 64, 0.125, p, 77, ptr(o8), ptr(sc), ptr(am), stream()), 'f8')
        w8 = (out.float() / sc).clamp(-448, 448).to(torch.float8_e4m3fn)
        ok = torch.equal(out, out2) and torch.equal(lse, lse2) and float(((o8.view(torch.float8_e4m3fn).float() - w8.float()).abs() > 0).float().mean()) < 1e-4
        r1, r2 = (torch.full((B * N, 3 * d), float('nan'), device='cuda', dtype=bf) for _ in range(2))
        check(lib().ecgvit_attention_bwd(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(r1), B, N, h, 64, 0.125, p, 77, hip.BF16, stream()), 'b')
        sc2 = (r1.float().abs().max() / 57344.0).reshape(1).clamp_min(1e-9); am.zero_()
        g8 = torch.zeros(B * N, 3 * d, dtype=torch.uint8, device='cuda')
        check(lib().ecgvit_attention_bwd_q8(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(r2), B, N, h, 64, 0.125, p, 77, ptr(g8), ptr(sc2), ptr(am), stream()), 'b8')
        wg = (r1.float() / sc2).clamp(-57344, 57344).to(torch.float8_e5m2)
        ok = ok and torch.equal(r1, r2) and float(am) == float(r1.float().abs().max()) and \
            float(((g8.view(torch.float8_e5m2).float() - wg.float()).abs() > 0).float().mean()) < 1e-4
        if not ok:
            bad += 1; print('ATTN8 MISMATCH', B, h, N, p, flush=True)
        n_at += 1
    else:
        h = rng.choice([1, 2, 3, 5, 12]); N = rng.randrange(129, 513); B = rng.choice([3, 40, 90, 300]) if h < 12 else rng.choice([8, 30, 64])
        p = rng.choice([0.0, 0.1, 0.3]); d = h * 64
        qkv = (torch.randn(B * N, 3 * d, device='cuda') * 1.2).to(bf); do = torch.randn(B * N, d, device='cuda').to(bf)
        out = torch.empty(B * N, d, device='cuda', dtype=bf); lse = torch.zeros(B * h * N, device='cuda')
        check(lib().ecgvit_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, h, 64, 0.125, p, 77, hip.BF16, stream()), 'f')
        res = []
        for pers in (True, True, True, N > 256):   # the one-item kernel covers N <= 256 only
            fn = lib().ecgvit_attention_bwd if pers else tools_lib().ecgvit_attention_bwd_oneitem
            r = torch.full((B * N, 3 * d), float('nan'), device='cuda', dtype=bf)
            check(fn(ptr(qkv), ptr(out), ptr(do), ptr(lse), ptr(r), B, N, h, 64, 0.125, p, 77, hip.BF16, stream()), 'b')
            torch.cuda.synchronize(); res.append(r)
        if not (torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])):
            bad += 1; print('ATTN NONDETERMINISTIC', B, h, N, p, flush=True)
        err = float((res[0].float() - res[3].float()).norm() / res[3].float().norm())
        if not (err < 3e-3) or not torch.isfinite(res[0].float()).all():
            bad += 1; print('ATTN MISMATCH vs one-item kernel', B, h, N, p, err, flush=True)
        n_at += 1
print(f'stress done: {n_nt} A.B^T shapes ({n_aux8} launches of the x-aux e4m3 form among them), {n_tn} A^T.B shapes, {n_at} attention cases, {bad} failures', flush=True)
sys.exit(1 if bad else 0)
