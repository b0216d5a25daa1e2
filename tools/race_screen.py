"""Race screen for the 256x256 GEMM's DMA schedule (sync-structure edits need one: a read placed one phase early passes
every single run in which the DMA happens to land first).  Each shape: many launches of the 256x256 kernel, bf16 and fp8,
with and without a second stream hammering HBM, every output compared bit for bit with the 128x128 kernel's (same arithmetic,
different pipeline)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
reps = int(os.environ.get("REPS", 60))
noise_src = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
noise_dst = torch.empty_like(noise_src)
side = torch.cuda.Stream()
bad, runs = [], 0
for (M, N, K) in [(256, 256, 128), (512, 512, 512), (1000, 520, 256), (2048, 4096, 4096), (4096, 1024, 16384), (8192, 8192, 1024), (777, 3000, 1152), (8192, 4096, 4096), (4096, 4096, 16384)]:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev, generator=g).to(BF16)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(BF16)
    for layout in ("ft", "rm"):
        lin = ops.PackedLinear(w, tiled=True, rowmajor=True)
        ref = ops.gemm(a, lin, layout=layout, tile=128, split_k=1, out_dtype=torch.float32)
        do8 = K % 256 == 0                  # the 256x256 fp8 kernel needs whole 128-pair K-tile pairs
        if do8:
            lin8 = ops.PackedLinearFP8(w, tiled=True, rowmajor=True)
            aq, asc = ops.quantize_rows_fp8(a)
            ref8 = ops.gemm_fp8(aq, asc, lin8, layout=layout, tile=128, split_k=1, out_dtype=torch.float32)
        for noisy in (False, True):
            for r in range(reps):
                if noisy and r % 4 == 0:
                    with torch.cuda.stream(side):
                        noise_dst.copy_(noise_src, non_blocking=True)
                o = ops.gemm(a, lin, layout=layout, tile=256, out_dtype=torch.float32)
                runs += 1
                if not torch.equal(o, ref): bad.append(("bf16", M, N, K, layout, noisy, r))
                if do8:
                    o8 = ops.gemm_fp8(aq, asc, lin8, layout=layout, tile=256, out_dtype=torch.float32)
                    runs += 1
                    if not torch.equal(o8, ref8): bad.append(("fp8", M, N, K, layout, noisy, r))
            torch.cuda.synchronize()
print(json.dumps({"kind": "race_screen", "launches_compared": runs, "mismatches": len(bad), "first": bad[:5]}))
