"""yardstick only: which hipBLASLt kernels torch.matmul picks for the train-step GEMM shapes (run under rocprofv3 --kernel-trace)"""
import torch
M = 512 * 251
bf = torch.bfloat16
for kin, nout in ((768, 2304), (768, 768), (768, 3072), (3072, 768)):
    X = torch.randn(M, kin, device='cuda').to(bf)
    W = torch.randn(nout, kin, device='cuda').to(bf)
    dY = torch.randn(M, nout, device='cuda').to(bf)
    for _ in range(5):
        torch.matmul(X, W.t())
        torch.matmul(dY, W)
torch.cuda.synchronize()
