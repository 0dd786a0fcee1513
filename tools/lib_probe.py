import torch
M,K,N=512*251,768,2304
X=torch.randn(M,K,device='cuda').bfloat16(); W=torch.randn(N,K,device='cuda').bfloat16(); C=torch.empty(M,N,device='cuda',dtype=torch.bfloat16)
for _ in range(3): torch.matmul(X,W.t(),out=C)
X2=torch.randn(M,3072,device='cuda').bfloat16(); W2=torch.randn(768,3072,device='cuda').bfloat16(); C2=torch.empty(M,768,device='cuda',dtype=torch.bfloat16)
for _ in range(3): torch.matmul(X2,W2.t(),out=C2)
torch.cuda.synchronize()
