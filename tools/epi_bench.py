#!/usr/bin/env python3
"""epilogue cost breakdown of the FFN-up forward GEMM (M=128512, K=768, N=3072) and FFN-down dgrad"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_representation_learning_amd import hip
M, d, f, bf = 512 * 251, 768, 3072, torch.bfloat16
X = torch.randn(M, d, device='cuda').to(bf); W1 = (torch.randn(f, d, device='cuda') * 0.02).to(bf)
H = torch.empty(M, f, device='cuda', dtype=bf); PRE = torch.empty(M, f, device='cuda', dtype=bf)
bias = torch.randn(f, device='cuda'); ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda'); cs = torch.empty(f, device='cuda')
dY = torch.randn(M, d, device='cuda').to(bf); W2 = (torch.randn(d, f, device='cuda') * 0.02).to(bf)
R = torch.randn(M, d, device='cuda').to(bf); Y = torch.empty(M, d, device='cuda', dtype=bf); bd = torch.randn(d, device='cuda')
def t(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
E = hip
W2t = W2.t().contiguous()
cases = {
 'up plain': lambda: hip.gemm(E.GEMM_NT, X, W1, H, M, f, d, d, d, f),
 'up bias': lambda: hip.gemm(E.GEMM_NT, X, W1, H, M, f, d, d, d, f, epilogue=E.EPI_BIAS, bias=bias),
 'up bias+gelu(+aux)': lambda: hip.gemm(E.GEMM_NT, X, W1, H, M, f, d, d, d, f, epilogue=E.EPI_BIAS | E.EPI_GELU, bias=bias, aux=PRE, ldaux=f),
 'up bias+gelu+drop': lambda: hip.gemm(E.GEMM_NT, X, W1, H, M, f, d, d, d, f, epilogue=E.EPI_BIAS | E.EPI_GELU | E.EPI_DROPOUT, bias=bias, aux=PRE, ldaux=f, dropout_p=0.1, seed=5),
 'up bias+gelu+gradaux+drop (shipped fwd)': lambda: hip.gemm(E.GEMM_NT, X, W1, H, M, f, d, d, d, f, epilogue=E.EPI_BIAS | E.EPI_GELU | E.EPI_GELU_GRAD_AUX | E.EPI_DROPOUT, bias=bias, aux=PRE, ldaux=f, dropout_p=0.1, seed=5),
 'dgradT mul_aux+colsum (shipped bwd)': lambda: hip.gemm(E.GEMM_NT, dY, W2t, H, M, f, d, d, d, f, epilogue=E.EPI_MUL_AUX | E.EPI_COLSUM, aux=PRE, ldaux=f, workspace=ws, colsum_out=cs),
 'dgradT mul_aux only': lambda: hip.gemm(E.GEMM_NT, dY, W2t, H, M, f, d, d, d, f, epilogue=E.EPI_MUL_AUX, aux=PRE, ldaux=f),
 'dgradT colsum only': lambda: hip.gemm(E.GEMM_NT, dY, W2t, H, M, f, d, d, d, f, epilogue=E.EPI_COLSUM, workspace=ws, colsum_out=cs),
 'dgrad plain': lambda: hip.gemm(E.GEMM_NN, dY, W2, H, M, f, d, d, f, f),
 'dgrad gelu_bwd': lambda: hip.gemm(E.GEMM_NN, dY, W2, H, M, f, d, d, f, f, epilogue=E.EPI_GELU_BWD, aux=PRE, ldaux=f),
 'dgrad gelu_bwd+drop': lambda: hip.gemm(E.GEMM_NN, dY, W2, H, M, f, d, d, f, f, epilogue=E.EPI_GELU_BWD | E.EPI_DROPOUT, aux=PRE, ldaux=f, dropout_p=0.1, seed=5),
 'dgrad gelu_bwd+drop+colsum': lambda: hip.gemm(E.GEMM_NN, dY, W2, H, M, f, d, d, f, f, epilogue=E.EPI_GELU_BWD | E.EPI_DROPOUT | E.EPI_COLSUM, aux=PRE, ldaux=f, dropout_p=0.1, seed=5, workspace=ws, colsum_out=cs),
 'dgradT plain': lambda: hip.gemm(E.GEMM_NT, dY, W2t, H, M, f, d, d, d, f),
 'dgradT gelu_bwd+drop+colsum': lambda: hip.gemm(E.GEMM_NT, dY, W2t, H, M, f, d, d, d, f, epilogue=E.EPI_GELU_BWD | E.EPI_DROPOUT | E.EPI_COLSUM, aux=PRE, ldaux=f, dropout_p=0.1, seed=5, workspace=ws, colsum_out=cs),
 'down plain': lambda: hip.gemm(E.GEMM_NT, H, W2, Y, M, d, f, f, f, d),
 'down bias+res+drop': lambda: hip.gemm(E.GEMM_NT, H, W2, Y, M, d, f, f, f, d, epilogue=E.EPI_BIAS | E.EPI_RESIDUAL | E.EPI_DROPOUT, bias=bd, residual=R, ldr=d, dropout_p=0.1, seed=7),
}
for k, fn in cases.items():
    print(f'{k:32s} {t(fn):8.1f} us', flush=True)
