import torch
A = torch.randn(4096, 1024, device="cuda"); B = torch.randn(1024, 1024, device="cuda"); C = torch.empty(4096, 1024, device="cuda")
for _ in range(5):
    torch.mm(A, B.t(), out=C)
    torch.mm(A, B, out=C)
    torch.mm(A.t()[:1024], A[:, :1024], out=C[:1024])
torch.cuda.synchronize()
