"""Which kernels does the vendor library pick for the training GEMM shapes?  Run under rocprofv3 --kernel-trace; the Tensile
kernel names spell out macro tile, matrix instruction, wave layout and LDS options."""
import torch
dev = torch.device("cuda:0")
for (M, N, K) in [(32768, 12288, 4096), (32768, 4096, 4096), (32768, 16384, 4096), (32768, 4096, 16384), (8192, 8192, 8192), (456, 28672, 4096)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
