"""Tiny driver for the rocprofv3 PMC passes: a handful of launches of the decode
weight-streaming GEMM at the fc_in shape (N=16384, K=4096, M=8; 134.2 MB of
weights per launch), rotating over 6 weight copies (> 256 MiB Infinity Cache)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N, K, M = 16384, 4096, 8
lins = [ops.PackedLinear((torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)) for _ in range(6)]
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for i in range(12):
    ops.gemm_skinny(x, lins[i % 6], out=out)
torch.cuda.synchronize()
print("done")
