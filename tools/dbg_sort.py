import torch
x = torch.rand(1024, device="cuda")
for _ in range(3):
    p = torch.argsort(x)
g = torch._standard_gamma(torch.full((2,), 0.9, device="cuda"))
torch.cuda.synchronize()
print(p[:4], g)
