import torch, time
for B, n in ((4, 256), (2, 512), (1, 1024), (512, 32)):
    A = torch.randn(B, n, n, device="cuda"); A = A + A.transpose(1, 2)
    torch.linalg.eigh(A); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): torch.linalg.eigh(A)
    torch.cuda.synchronize()
    print(B, n, "eigh ms", (time.perf_counter() - t0) / 5 * 1e3)
