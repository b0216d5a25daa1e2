"""Two launches each of the hand-written 256x256 GEMM and of torch.matmul (hipBLASLt) on the training shapes -- target of
tools/gpu_pmc_gemm_util.sh (rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
for (M, N, K) in [(32768, 4096, 4096), (32768, 16384, 4096), (32768, 4096, 16384)]:
    a = torch.randn(M, K, device=dev).to(BF16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
    lin = ops.PackedLinear(w)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    for _ in range(2):
        ops.gemm(a, lin, out=out, tile=256)
        torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
