"""Kernel micro-benchmarks on one MI355X (run through gpurun).  Random data only
(zero-filled operands inflate MFMA clocks, guide 5.4 rule 25).  Weight-streaming
kernels rotate over enough distinct weight copies to defeat the 256 MiB
Infinity Cache.  Writes JSON lines to gpurun_out/kbench.jsonl."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
OUT = os.path.join("gpurun_out", "kbench.jsonl")
os.makedirs("gpurun_out", exist_ok=True)
fout = open(OUT, "a")


def emit(**kw):
    line = json.dumps(kw)
    print(line, flush=True)
    fout.write(line + "\n")
    fout.flush()


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def bench_gemm(M, N, K, layout, act=0, tag=""):
    a = torch.randn(M, K, device=dev).to(BF16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
    lin = ops.PackedLinear(w, tiled=(layout == "ft"), rowmajor=(layout == "rm"))
    out = torch.empty(M, N, dtype=BF16, device=dev)
    for tile in (128, 256, 257):
        if tile >= 256 and K % 128:
            continue
        ms = timeit(lambda i: ops.gemm(a, lin, out=out, layout=layout, act=act, tile=tile), 20)
        emit(kind="gemm", tag=tag, M=M, N=N, K=K, layout=layout, tile=tile, ms=ms, tflops=2.0 * M * N * K / ms / 1e9)


def bench_prefill(M, N, K, tag):
    nbytes = N * K * 2
    ncopy = max(2, int(600e6 // nbytes) + 1)
    lins = [ops.PackedLinear((torch.randn(N, K, device=dev) * 0.05).to(BF16)) for _ in range(ncopy)]
    a = torch.randn(M, K, device=dev).to(BF16)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    for sk in (1, 0, 2, 3, 4, 6, 8, 16):
        try:
            ms = timeit(lambda i: ops.gemm(a, lins[i % ncopy], out=out, act=1, split_k=sk), 4 * ncopy, warmup=ncopy)
        except Exception as e:  # noqa: BLE001
            emit(kind="prefill_gemm", tag=tag, split_k=sk, error=str(e)[:120])
            continue
        emit(kind="prefill_gemm", tag=tag, M=M, N=N, K=K, split_k=sk, ms=ms, tflops=2.0 * M * N * K / ms / 1e9,
             gbps=nbytes / ms / 1e6)


def bench_skinny(M, N, K, variants, tag=""):
    nbytes = N * K * 2
    ncopy = max(2, int(600e6 // nbytes) + 1)
    lins = []
    for c in range(ncopy):
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        lins.append(ops.PackedLinear(w))
        del w
    x = torch.randn(M, K, device=dev).to(BF16)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    ksteps = lins[0].Kp // 32
    for (nt, waves, kc) in variants:
        if ksteps % (waves * kc):
            continue
        v = nt | waves << 4 | kc << 8
        try:
            ms = timeit(lambda i: ops.gemm_skinny(x, lins[i % ncopy], out=out, variant=v), 4 * ncopy, warmup=ncopy)
        except Exception as e:  # noqa: BLE001
            emit(kind="skinny", tag=tag, M=M, N=N, K=K, nt=nt, waves=waves, kc=kc, error=str(e)[:200])
            continue
        emit(kind="skinny", tag=tag, M=M, N=N, K=K, nt=nt, waves=waves, kc=kc, ms=ms, gbps=nbytes / ms / 1e6)
    del lins


def bench_skinny_dma(M, N, K, tag=""):
    """the LDS-DMA loader-wave GEMV (nt_hint bit 17; bit 18 = non-temporal DMA) against the default register-direct GEMV"""
    nbytes = N * K * 2
    ncopy = max(2, int(600e6 // nbytes) + 1)
    lins = []
    for c in range(ncopy):
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        lins.append(ops.PackedLinear(w))
        del w
    x = torch.randn(M, K, device=dev).to(BF16)
    ref = torch.empty(M, N, dtype=BF16, device=dev)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm_skinny(x, lins[0], out=ref, variant=0)
    # nt_hint: bit 17 = LDS-DMA GEMV, 18 = non-temporal DMA, 19..21 = form (gemv_dma.hip), 22..30 = workgroups (0 = 256)
    D, NT = 1 << 17, 1 << 18
    for name, v in (("default", 0), ("dma", D), ("dma_nt", D | NT), ("dma2_xl2", D | 1 << 19), ("dma2_xl2_nt", D | NT | 1 << 19),
                    ("dma_nt_ring5", D | NT | 2 << 19), ("dma2_nt_ring5", D | NT | 3 << 19)):
        try:
            out.zero_()
            ops.gemm_skinny(x, lins[0], out=out, variant=v)
            torch.cuda.synchronize()
            err = float((out.float() - ref.float()).abs().max())
            ms = timeit(lambda i: ops.gemm_skinny(x, lins[i % ncopy], out=out, variant=v), 4 * ncopy, warmup=ncopy)
        except Exception as e:  # noqa: BLE001
            emit(kind="skinny_dma", tag=tag, variant=name, error=str(e)[:200])
            continue
        emit(kind="skinny_dma", tag=tag, M=M, N=N, K=K, variant=name, ms=ms, gbps=nbytes / ms / 1e6, max_abs_vs_default=err,
             ref_absmax=float(ref.float().abs().max()))
    del lins


def main():
    t0 = time.time()
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "gemm"):
        for layout in ("ft", "rm"):
            bench_gemm(4096, 4096, 4096, layout, tag="square4k")
            bench_gemm(8192, 8192, 8192, layout, tag="square8k")
            for (N, K, tag) in [(12288, 4096, "qkv"), (16384, 4096, "fc_in"), (4096, 4096, "out_proj"),
                                (4096, 16384, "fc_out"), (1024, 4096, "adapter_dn"), (4096, 1024, "adapter_up")]:
                bench_gemm(1216, N, K, layout, tag="prefill152_" + tag)
            bench_gemm(32768, 16384, 4096, layout, tag="train_fc_in")
    if which == "ksweep":   # per-tile overhead (a) vs per-K-tile cost (b) of the 256x256 kernel; clock droop on long runs
        for K in (256, 1024, 4096):
            a = torch.randn(8192, K, device=dev).to(BF16)
            lin = ops.PackedLinear((torch.randn(8192, K, device=dev) * 0.05).to(BF16))
            out = torch.empty(8192, 8192, dtype=BF16, device=dev)
            for tile in (256, 267, 268, 270):      # (MAGMA_HIP_LIB=magma_amd/libmagma_hip_abl.so, built with `make ABL=1`) 267: no epilogue; 268: LDS staging only; 270: all tiles store to tile (0,0)
                ms = timeit(lambda i: ops.gemm(a, lin, out=out, tile=tile, split_k=1), 10)
                emit(kind="ksweep", M=8192, N=8192, K=K, tile=tile, ms=ms, tflops=2.0 * 8192 * 8192 * K / ms / 1e9)
        a = torch.randn(32768, 4096, device=dev).to(BF16)
        lin = ops.PackedLinear((torch.randn(16384, 4096, device=dev) * 0.05).to(BF16))
        out = torch.empty(32768, 16384, dtype=BF16, device=dev)
        for iters in (1, 3, 10, 40):
            ms = timeit(lambda i: ops.gemm(a, lin, out=out, tile=256), iters, warmup=1)
            emit(kind="droop", M=32768, N=16384, K=4096, iters=iters, ms=ms, tflops=2.0 * 32768 * 16384 * 4096 / ms / 1e9)
    if which == "fp8":   # fp8 (MX-rate MFMA) vs bf16 on the config-5 GEMMs: QKV, out_proj, adapters (and fc for reference)
        for M in (456, 32768):
            for (N, K, tag) in [(12288, 4096, "qkv"), (4096, 4096, "out_proj"), (1024, 4096, "adapter_dn"), (4096, 1024, "adapter_up"),
                                (16384, 4096, "fc_in")]:
                a = torch.randn(M, K, device=dev).to(BF16)
                w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
                lin, lin8 = ops.PackedLinear(w), ops.PackedLinearFP8(w)
                out = torch.empty(M, N, dtype=BF16, device=dev)
                aq, asc = ops.quantize_rows_fp8(a)
                it = 20 if M < 4096 else 5
                ms16 = timeit(lambda i: ops.gemm(a, lin, out=out), it)
                ms8 = timeit(lambda i: ops.gemm_fp8(aq, asc, lin8, out=out), it)
                msq = timeit(lambda i: ops.quantize_rows_fp8(a), it)
                fl = 2.0 * M * N * K
                emit(kind="fp8", tag=tag, M=M, N=N, K=K, bf16_ms=ms16, fp8_ms=ms8, quant_ms=msq, bf16_tflops=fl / ms16 / 1e9,
                     fp8_tflops=fl / ms8 / 1e9)
    if which == "fp8tile":   # fp8 GEMM: 128x128 kernel vs 256x256 kernel vs bf16 256x256, training shapes, with a numerics check
        M = 32768
        for (N, K, tag) in [(12288, 4096, "qkv"), (4096, 4096, "out_proj"), (16384, 4096, "fc_in"), (4096, 16384, "fc_out")]:
            a = torch.randn(M, K, device=dev).to(BF16)
            w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
            lin, lin8 = ops.PackedLinear(w), ops.PackedLinearFP8(w)
            aq, asc = ops.quantize_rows_fp8(a)
            o128 = ops.gemm_fp8(aq, asc, lin8, tile=128)
            o256 = ops.gemm_fp8(aq, asc, lin8, tile=256)
            diff = float((o128.float() - o256.float()).abs().max()), float(o128.float().abs().max())
            out = torch.empty(M, N, dtype=BF16, device=dev)
            fl = 2.0 * M * N * K
            r = {"kind": "fp8tile", "tag": tag, "M": M, "N": N, "K": K, "max_abs_diff_128_vs_256": diff[0], "max_abs": diff[1]}
            for name, fn in (("bf16_256", lambda i: ops.gemm(a, lin, out=out)), ("fp8_128", lambda i: ops.gemm_fp8(aq, asc, lin8, out=out, tile=128)),
                             ("fp8_256", lambda i: ops.gemm_fp8(aq, asc, lin8, out=out, tile=256))):
                ms = timeit(fn, 5)
                r[name + "_ms"] = ms; r[name + "_tflops"] = fl / ms / 1e9
            emit(**r)
    if which == "mfma":   # 256x256 bf16 kernel: 16x16x32 MFMA (tile 259) vs 32x32x16 (tile 258), interleaved A/B, random data
        shapes = [(32768, 12288, 4096, "qkv"), (32768, 4096, 4096, "out_proj"), (32768, 16384, 4096, "fc_in"),
                  (32768, 4096, 16384, "fc_out"), (32768, 1024, 4096, "adapter_dn"), (32768, 4096, 1024, "adapter_up"),
                  (4096, 4096, 4096, "square4k"), (8192, 8192, 8192, "square8k")]
        for (M, N, K, tag) in shapes:
            a = torch.randn(M, K, device=dev).to(BF16)
            w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
            lin = ops.PackedLinear(w)
            out = torch.empty(M, N, dtype=BF16, device=dev)
            o16 = ops.gemm(a, lin, tile=259).float()
            o32 = ops.gemm(a, lin, tile=258).float()
            r = {"kind": "mfma", "tag": tag, "M": M, "N": N, "K": K,
                 "rel_diff_32_vs_16": float((o32 - o16).norm() / o16.norm())}
            del o16, o32
            best = {259: 1e9, 258: 1e9}
            for rep in range(3):                       # interleaved: both arms see the same clocks / thermals
                for tile in (259, 258):
                    best[tile] = min(best[tile], timeit(lambda i: ops.gemm(a, lin, out=out, tile=tile), 6, warmup=2))
            fl = 2.0 * M * N * K
            r.update(mfma16_ms=best[259], mfma32_ms=best[258], mfma16_tflops=fl / best[259] / 1e9, mfma32_tflops=fl / best[258] / 1e9,
                     speedup=best[259] / best[258])
            emit(**r)
    if which == "mx":   # fp8 GEMM at 5 PF dense peak: MX block-scaled vs per-row scaled (128x128, 256x256 kernels) + the quantisers
        for M in (456, 32768):
            for (N, K, tag) in [(12288, 4096, "qkv"), (4096, 4096, "out_proj"), (16384, 4096, "fc_in"), (4096, 16384, "fc_out")]:
                a = torch.randn(M, K, device=dev).to(BF16)
                w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
                lin8, linx = ops.PackedLinearFP8(w), ops.PackedLinearMX(w)
                out = torch.empty(M, N, dtype=BF16, device=dev)
                aq, asc = ops.quantize_rows_fp8(a)
                xq, xsc = ops.quantize_mx_fp8(a)
                it = 20 if M < 4096 else 5
                fl = 2.0 * M * N * K
                r = {"kind": "mx", "tag": tag, "M": M, "N": N, "K": K}
                for name, fn in (("row_128", lambda i: ops.gemm_fp8(aq, asc, lin8, out=out, tile=128)),
                                 ("row_256", lambda i: ops.gemm_fp8(aq, asc, lin8, out=out, tile=256 if M >= 256 else 128)),
                                 ("mx_128", lambda i: ops.gemm_mx_fp8(xq, xsc, linx, out=out, tile=128)),
                                 ("mx_256", lambda i: ops.gemm_mx_fp8(xq, xsc, linx, out=out, tile=256 if M >= 256 else 128)),
                                 ("quant_row", lambda i: ops.quantize_rows_fp8(a)), ("quant_mx", lambda i: ops.quantize_mx_fp8(a))):
                    ms = timeit(fn, it)
                    r[name + "_ms"] = ms
                    if not name.startswith("quant"):
                        r[name + "_tflops"] = fl / ms / 1e9
                        r[name + "_frac_of_5PF"] = fl / ms / 1e9 / 5000.0
                emit(**r)
    if which == "q8":   # the fp8 256x256 GEMM with and without the MX e4m3 copy written by its epilogue (mg_epilogue.C8)
        M, N, K = 32768, 16384, 4096
        a = torch.randn(M, K, device=dev).to(BF16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        lin8 = ops.PackedLinearFP8(w)
        aq, asc = ops.quantize_rows_fp8(a)
        out = torch.empty(M, N, dtype=BF16, device=dev)
        pre = torch.empty(M, N, dtype=BF16, device=dev)
        aux = torch.randn(M, N, device=dev).to(BF16)
        q, sc = ops.mx_empty(M, N, dev)
        fl = 2.0 * M * N * K
        for name, fn in (("fwd_bf16_out", lambda i: ops.gemm_fp8(aq, asc, lin8, out=out, act=1, out2=pre)),
                         ("fwd_mx_only", lambda i: ops.gemm_fp8(aq, asc, lin8, act=1, out2=pre, mx_out=(q, sc), no_out=True)),
                         ("fwd_mx_and_bf16", lambda i: ops.gemm_fp8(aq, asc, lin8, out=out, act=1, out2=pre, mx_out=(q, sc))),
                         ("dgrad_bf16_out", lambda i: ops.gemm_fp8(aq, asc, lin8, out=out, aux=aux, aux_mode=ops.MG_AUX_GELU_GRAD)),
                         ("dgrad_mx_only", lambda i: ops.gemm_fp8(aq, asc, lin8, aux=aux, aux_mode=ops.MG_AUX_GELU_GRAD, mx_out=(q, sc), no_out=True)),
                         ("plain_bf16_out", lambda i: ops.gemm_fp8(aq, asc, lin8, out=out)),
                         ("plain_mx_only", lambda i: ops.gemm_fp8(aq, asc, lin8, mx_out=(q, sc), no_out=True)),
                         ("quantize_mx_pass", lambda i: ops.quantize_mx_fp8(out)), ("quantize_rows_pass", lambda i: ops.quantize_rows_fp8(out))):
            ms = timeit(fn, 5)
            emit(kind="q8", case=name, M=M, N=N, K=K, ms=ms, tflops=fl / ms / 1e9 if "quantize" not in name else None)
    if which == "lib":   # calibration: the vendor libraries on the same shapes (torch.matmul -> hipBLASLt / rocBLAS, SDPA)
        for (M, N, K, tag) in [(32768, 12288, 4096, "qkv"), (32768, 4096, 4096, "out_proj"), (32768, 16384, 4096, "fc_in"),
                               (32768, 4096, 16384, "fc_out"), (8192, 8192, 8192, "square8k"), (456, 28672, 4096, "prefill_qkv_fc_in")]:
            a = torch.randn(M, K, device=dev).to(BF16)
            w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
            lin = ops.PackedLinear(w)
            out = torch.empty(M, N, dtype=BF16, device=dev)
            wt = w.t()
            fl = 2.0 * M * N * K
            r = {"kind": "lib_gemm", "tag": tag, "M": M, "N": N, "K": K}
            for name, fn in (("mine", lambda i: ops.gemm(a, lin, out=out, tile=256 if M <= 512 else 0)),
                             ("torch_nt", lambda i: torch.matmul(a, wt, out=out)),
                             ("mine_again", lambda i: ops.gemm(a, lin, out=out, tile=256 if M <= 512 else 0)),
                             ("torch_nt_again", lambda i: torch.matmul(a, wt, out=out))):
                ms = timeit(fn, 10)
                r[name + "_ms"] = ms
                r[name + "_tflops"] = fl / ms / 1e9
            emit(**r)
        import torch.nn.functional as Fn
        B, H, S = 16, 16, 2048
        q = (torch.randn(B, H, S, 256, device=dev) * 0.5).to(BF16)
        k = (torch.randn(B, H, S, 256, device=dev) * 0.5).to(BF16)
        v = torch.randn(B, H, S, 256, device=dev).to(BF16)
        fl = B * H * 4.0 * S * S * 256 / 2
        r = {"kind": "lib_attention", "B": B, "H": H, "S": S, "dh": 256}
        try:
            ms = timeit(lambda i: Fn.scaled_dot_product_attention(q, k, v, is_causal=True, scale=1.0 / 16), 5)
            r["sdpa_fwd_ms"] = ms
            r["sdpa_fwd_tflops_causal"] = fl / ms / 1e9
            qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))
            o = Fn.scaled_dot_product_attention(qg, kg, vg, is_causal=True, scale=1.0 / 16)
            do = torch.randn_like(o)
            ms = timeit(lambda i: torch.autograd.grad(o, (qg, kg, vg), do, retain_graph=True), 5)
            r["sdpa_bwd_ms"] = ms
        except Exception as e:  # noqa: BLE001
            r["sdpa_error"] = str(e)[:300]
        hs = H * S * 256
        vt = ops.head_transpose(v, B, H, S, sb=hs, ss=256, sh=S * 256)
        out = torch.empty(B * S, H * 256, dtype=BF16, device=dev)
        lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
        ms = timeit(lambda i: ops.attn_prefill(q, k, vt, out, B, H, S, lse=lse), 5)
        r["mine_fwd_ms"] = ms
        r["mine_fwd_tflops_causal"] = fl / ms / 1e9
        qt = ops.head_transpose(q, B, H, S, sb=hs, ss=256, sh=S * 256)
        kt = ops.head_transpose(k, B, H, S, sb=hs, ss=256, sh=S * 256)
        dO = torch.randn(B * S, H * 256, device=dev).to(BF16)
        dOt = ops.head_transpose(dO, B, H, S, sb=S * H * 256, ss=H * 256, sh=256)
        ms = timeit(lambda i: ops.attn_bwd(q, k, v, qt, kt, dO, dOt, out, lse, B, H, S), 5)
        r["mine_bwd_ms"] = ms
        emit(**r)
    if which == "abl":   # timing ablations of the 256x256 bf16 kernel (results of 261..264 are WRONG by construction)
        for (M, N, K, tag) in [(32768, 16384, 4096, "fc_in"), (32768, 4096, 16384, "fc_out"), (32768, 4096, 4096, "out_proj"), (32768, 12288, 4096, "qkv"),
                               (8192, 8192, 8192, "square8k")]:
            a = torch.randn(M, K, device=dev).to(BF16)
            lin = ops.PackedLinear((torch.randn(N, K, device=dev) * 0.05).to(BF16))
            out = torch.empty(M, N, dtype=BF16, device=dev)
            fl = 2.0 * M * N * K
            ref = ops.gemm(a, lin, tile=256).float()
            old = ops.gemm(a, lin, tile=266)
            r = {"kind": "abl", "equal_to_first_schedule": bool(torch.equal(old.float(), ref)), "tag": tag, "M": M, "N": N, "K": K}
            for rep in range(2):
                for name, tile in (("full", 256), ("l2hot_dma", 261), ("no_ds_read", 262), ("no_mfma", 263), ("no_dma", 264), ("first_dma_schedule", 266), ("balanced_fragment_reads", 272)):
                    ms = timeit(lambda i: ops.gemm(a, lin, out=out, tile=tile), 8)
                    r[f"{name}_ms_{rep}"] = round(ms, 4)
                    r[f"{name}_tf_{rep}"] = round(fl / ms / 1e9, 1)
            emit(**r)
    if which == "pad":   # does a power-of-two row stride of A cost bandwidth (L2 / HBM channel aliasing)?  lda = K + pad elements
        for (M, N, K, tag) in [(32768, 4096, 16384, "fc_out"), (32768, 4096, 4096, "out_proj"), (32768, 16384, 4096, "fc_in"), (8192, 8192, 8192, "square8k")]:
            lin = ops.PackedLinear((torch.randn(N, K, device=dev) * 0.05).to(BF16))
            out = torch.empty(M, N, dtype=BF16, device=dev)
            fl = 2.0 * M * N * K
            r = {"kind": "pad", "tag": tag, "M": M, "N": N, "K": K}
            for rep in range(2):
                for pad in (0, 64, 128, 256, 576):
                    a = torch.randn(M, K + pad, device=dev).to(BF16)[:, :K]
                    ms = timeit(lambda i: ops.gemm(a, lin, out=out, tile=256), 8)
                    r[f"pad{pad}_tf_{rep}"] = round(fl / ms / 1e9, 1)
                    del a
            emit(**r)
    if which == "group":   # tile-walk block shape of the 256x256 kernel: group_m row-tiles x 32/group_m column-tiles per XCD at a time
        for (M, N, K, tag) in [(32768, 4096, 16384, "fc_out"), (32768, 4096, 4096, "out_proj"), (32768, 16384, 4096, "fc_in"), (32768, 12288, 4096, "qkv"),
                               (8192, 8192, 8192, "square8k"), (4096, 16384, 32768, "wgrad_fc_in")]:
            a = torch.randn(M, K, device=dev).to(BF16)
            lin = ops.PackedLinear((torch.randn(N, K, device=dev) * 0.05).to(BF16))
            out = torch.empty(M, N, dtype=BF16, device=dev)
            fl = 2.0 * M * N * K
            r = {"kind": "group", "tag": tag, "M": M, "N": N, "K": K}
            for rep in range(2):
                for gm in (8, 1, 2, 4, 16, 32):
                    os.environ["MAGMA_G256_GROUP_M"] = str(gm)
                    ms = timeit(lambda i: ops.gemm(a, lin, out=out, tile=256), 8)
                    r[f"gm{gm}_tf_{rep}"] = round(fl / ms / 1e9, 1)
            os.environ.pop("MAGMA_G256_GROUP_M", None)
            emit(**r)
    if which == "tilecost":   # per-tile overhead a and per-K-tile cost b of the 256x256 kernel (T_tile = a + b * K/64) for three epilogues
        M = N = 8192
        r1 = torch.randn(M, N, device=dev).to(BF16)
        aux = torch.randn(M, N, device=dev).to(BF16)
        out = torch.empty(M, N, dtype=BF16, device=dev)
        for name, kw in (("plain", {}), ("bias_gelu", {"act": ops.MG_ACT_GELU_NEW}), ("one_residual", {"residuals": (r1,)}),
                         ("gelu_grad_aux", {"aux": aux, "aux_mode": ops.MG_AUX_GELU_GRAD})):
            pts = {}
            for K in (1024, 4096, 16384):
                a = torch.randn(M, K, device=dev).to(BF16)
                lin = ops.PackedLinear((torch.randn(N, K, device=dev) * 0.05).to(BF16), bias=torch.randn(N, device=dev))
                try:
                    ms = min(timeit(lambda i: ops.gemm(a, lin, out=out, tile=256, **kw), 8) for _ in range(2))
                except TypeError as e:
                    pts = {"error": str(e)[:100]}
                    break
                pts[K] = ms * 1e3 / 4          # us per tile (1024 tiles on 256 CUs)
            if "error" not in pts:
                b = (pts[16384] - pts[4096]) / 192
                emit(kind="tilecost", epilogue=name, us_per_tile=pts, b_us_per_ktile=b, a_us_per_tile=pts[4096] - 64 * b,
                     a_from_1024=pts[1024] - 16 * b)
            else:
                emit(kind="tilecost", epilogue=name, **pts)
    if which == "stamps":   # (make ABL=1) in-kernel 100-MHz time stamps of the 256x256 kernel: where does a tile's time go?
        for (M, N, K, tag) in [(8192, 8192, 4096, "square_k4096"), (32768, 12288, 4096, "qkv"), (32768, 4096, 4096, "out_proj"), (32768, 16384, 4096, "fc_in"),
                               (32768, 4096, 16384, "fc_out"),
                               (8192, 8192, 1024, "square_k1024")]:
            a = torch.randn(M, K, device=dev).to(BF16)
            lin = ops.PackedLinear((torch.randn(N, K, device=dev) * 0.05).to(BF16))
            out = torch.empty(M, N, dtype=BF16, device=dev)
            nt = (M // 256) * (N // 256)
            ws = ops.splitk_workspace(dev)
            for _ in range(3):
                ops.gemm(a, lin, out=out, tile=271)
            torch.cuda.synchronize()
            st = ws[: nt * 16].view(torch.int64).view(nt, 8).cpu()
            t = st[:, :7].double() / 100.0                       # us
            seg = (t[:, 1:] - t[:, :-1])
            names = ["k_loop(+prologue)", "stage0", "walk0", "stage1", "walk1", "drain"]
            r = {"kind": "stamps", "tag": tag, "M": M, "N": N, "K": K, "tiles": nt,
                 "mean_us": {n: round(float(seg[:, i].mean()), 2) for i, n in enumerate(names)},
                 "p90_us": {n: round(float(seg[:, i].quantile(0.9)), 2) for i, n in enumerate(names)},
                 "launch_span_us": round(float(t[:, 6].max() - t[:, 0].min()), 1)}
            # turnaround of a CU slot: workgroup j (by start time, beyond the first wave) starts when the (j - first_wave)-th
            # workgroup to finish has freed its CU; first_wave = workgroups that started within 20 us of the launch
            order = t[:, 0].argsort()
            starts = t[order, 0]
            ends = t[:, 6].sort().values
            first_wave = int((starts < starts[0] + 20.0).sum())
            gaps = (starts[first_wave:] - ends[: nt - first_wave]) if nt > first_wave else torch.zeros(1, dtype=torch.float64)
            r["first_wave"] = first_wave
            r["slot_turnaround_us"] = {"mean": round(float(gaps.mean()), 2), "p10": round(float(gaps.quantile(0.1)), 2), "p50": round(float(gaps.quantile(0.5)), 2), "p90": round(float(gaps.quantile(0.9)), 2)}
            r["first_wave_start_spread_us"] = round(float(starts[first_wave - 1] - starts[0]), 2)
            # effective shader clock under this kernel's load: s_memtime cycles of a workgroup's life / its wall time
            clk = st[:, 7].double() / (t[:, 6] - t[:, 0])        # cycles per us = MHz
            r["shader_clock_MHz"] = {"mean": round(float(clk.mean()), 1), "p10": round(float(clk.quantile(0.1)), 1), "p90": round(float(clk.quantile(0.9)), 1)}
            r["peak_at_this_clock_TF"] = round(2500.0 * float(clk.mean()) / 2400.0, 1)
            emit(**r)
    if which == "shortk":   # adapter up-projection (K = 1024) with three residual reads: epilogue-bound; 128x128 vs 256x256 kernel
        M, N, K = 32768, 4096, 1024
        a = torch.randn(M, K, device=dev).to(BF16)
        w = ops.RawWeight((torch.randn(N, K, device=dev) * 0.05).to(BF16), bias=torch.randn(N, device=dev))
        r1, r2, r3 = (torch.randn(M, N, device=dev).to(BF16) for _ in range(3))
        out = torch.empty(M, N, dtype=BF16, device=dev)
        for nres in (3, 1, 0):
            for tile in (256, 128):
                res = (r1, r2, r3)[:nres]
                ms = timeit(lambda i: ops.gemm(a, w, out=out, layout="rm", residuals=res, tile=tile), 10)
                emit(kind="shortk", M=M, N=N, K=K, residuals=nres, tile=tile, ms=ms, tflops=2.0 * M * N * K / ms / 1e9,
                     gbps=(M * N * 2 * (1 + nres) + M * K * 2) / ms / 1e6)
    if which == "wgrad":   # adapter weight-gradient shapes of the training step: the contraction runs over M = B*S = 32768 rows
        for (M, N, K) in [(4096, 1024, 32768), (1024, 4096, 32768)]:
            a = torch.randn(M, K, device=dev).to(BF16)
            w = ops.RawWeight((torch.randn(N, K, device=dev) * 0.05).to(BF16))
            out = torch.empty(M, N, dtype=torch.float32, device=dev)
            ref = None
            for sk in (1, 0, 2, 4, 8):
                ms = timeit(lambda i: ops.gemm(a, w, out=out, out_dtype=torch.float32, layout="rm", use_bias=False, split_k=sk), 10)
                if ref is None:
                    ref = out.clone()
                emit(kind="wgrad", M=M, N=N, K=K, split_k=sk, ms=ms, tflops=2.0 * M * N * K / ms / 1e9,
                     rel_vs_split1=float((out - ref).norm() / ref.norm()))
    if which == "prefill":   # M = 8 x 57 rows: weights rotate so they stream from HBM as in a real prefill
        for (N, K, tag) in [(12288, 4096, "qkv"), (16384, 4096, "fc_in"), (4096, 4096, "out_proj"),
                            (4096, 16384, "fc_out"), (1024, 4096, "adapter_dn"), (4096, 1024, "adapter_up")]:
            bench_prefill(456, N, K, tag)
    if which == "dma":
        for (N, K, tag) in [(28672, 4096, "qkv|fc_in"), (16384, 4096, "fc_in"), (4096, 4096, "out_proj"),
                            (1024, 4096, "adapter_dn"), (50272, 4096, "lm_head")]:
            bench_skinny_dma(8, N, K, tag=tag)
    if which == "skinny_cat":      # the [W_out | W_up] launch of the MAGMA_v1 decode block (K = 5120) and the adapter's down-projection
        variants = [(1, 8, 4), (1, 8, 2), (1, 8, 1), (1, 16, 2), (1, 16, 1), (1, 8, 5), (1, 8, 10), (1, 4, 8), (1, 4, 4), (2, 8, 4), (1, 16, 5), (1, 16, 10)]
        for (N, K, tag) in [(4096, 5120, "out|up"), (1024, 4096, "adapter_dn"), (4096, 4096, "out_proj")]:
            bench_skinny(8, N, K, variants, tag=tag)
    if which in ("all", "skinny"):
        variants = [(1, 8, 16), (2, 8, 16), (1, 4, 16), (2, 4, 16), (1, 8, 8), (2, 8, 8), (4, 8, 8), (2, 4, 8),
                    (4, 4, 8), (1, 8, 4), (2, 8, 4), (4, 8, 4), (1, 16, 8), (1, 16, 4), (1, 16, 2), (2, 16, 4)]
        for (N, K, tag) in [(12288, 4096, "qkv"), (16384, 4096, "fc_in"), (4096, 4096, "out_proj"),
                            (4096, 16384, "fc_out"), (1024, 4096, "adapter_dn"), (4096, 1024, "adapter_up"),
                            (50258, 4096, "lm_head")]:
            bench_skinny(8, N, K, variants, tag=tag)
    emit(kind="done", seconds=time.time() - t0)


if __name__ == "__main__":
    main()
