"""Adapter weight-gradient GEMMs of the training step in isolation (4096 x 1024 and 1024 x 4096 outputs over K = B*S = 32768,
row-major operands, fp32 out): 128x128 kernel with its automatic 2-way split vs the 256x256 kernel with split-K (automatic)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import ops
dev = torch.device("cuda:0")
BF = torch.bfloat16
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M, N in ((4096, 1024), (1024, 4096)):
    K = 32768
    a = torch.randn(M, K, device=dev).to(BF)
    w = ops.RawWeight((torch.randn(N, K, device=dev) * 0.05).to(BF))
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    r = {"M": M, "N": N, "K": K}
    for name, kw in (("tile128_auto_split", dict(tile=128)), ("auto", dict()), ("tile256_split2", dict(tile=256, split_k=2)), ("tile256_split4", dict(tile=256, split_k=4))):
        ms = t(lambda: ops.gemm(a, w, out=out, layout="rm", use_bias=False, **kw))
        r[name + "_us"] = ms * 1e3
        r[name + "_tflops"] = 2.0 * M * N * K / ms / 1e9
    print(json.dumps(r))
