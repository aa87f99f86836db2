"""Yardstick: what does the vendor fp32 GEMM (rocBLAS/hipBLASLt through torch.mm) reach on the GEMM shapes of this path?"""
import time
import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda")
shapes = [("r50 res2 2c  C64->K256 @56 b64", 256, 64, 200704), ("r50 res3 2c C128->K512 @28", 512, 128, 50176), ("r50 res4 2c C256->K1024 @14", 1024, 256, 12544),
          ("r50 res4 2a C1024->K256 @14", 256, 1024, 12544), ("r50 res5 2c C512->K2048 @7", 2048, 512, 3136), ("r50 res5 2a C2048->K512 @7", 512, 2048, 3136),
          ("mb pw C512->K512 @14 b256", 512, 512, 50176), ("wino xi-GEMM C256 K256 P3200 (x64 batched)", 256, 256, 3200), ("wino C512 K512 P800 (x64)", 512, 512, 800),
          ("wino C64 K64 P46208 (x64)", 64, 64, 46208)]
for name, M, K, N in shapes:
    batched = "x64" in name
    if batched:
        a = torch.rand((64, M, K), device=dev); b = torch.rand((64, K, N), device=dev)
        f = lambda: torch.bmm(a, b)
        flops = 2.0 * 64 * M * K * N
    else:
        a = torch.rand((M, K), device=dev); b = torch.rand((K, N), device=dev)
        f = lambda: torch.mm(a, b)
        flops = 2.0 * M * K * N
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    print(f"{name:48s} {dt*1e3:8.4f} ms  {flops/dt/1e12:7.1f} TF  {flops/dt/1e12/157.3*100:5.1f}%")
