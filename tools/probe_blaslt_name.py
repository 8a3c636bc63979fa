import torch
for (M, N, K) in [(786432, 768, 768), (786432, 2304, 768), (786432, 768, 3072)]:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): torch.matmul(x, w.t(), out=y)
    torch.cuda.synchronize()
