#!/usr/bin/env python
"""bench.py -- MAGMA_v1 on MI355X: generate tokens/sec (BASELINE.json config[1]:
bf16 inference, batch-8 images, 32 generated tokens) on synthetic data.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...     (--gpus must equal the rank count)

One "step" = one full pass of the inference hot path over one batch: CLIP
RN50x16 trunk + ImagePrefix on 8 images, word embeddings of an 8-token prompt,
GPT-J prefill (S0 = P + 8), 32 greedy decode steps (early stop disabled).
Inputs and weights are resident in HBM before the timed region.  Multi-GPU =
independent replicas (the inference path has no exchange step): value is the
sum over ranks / max-over-ranks time, scaling "weak".

Extra objects on the JSON line:
  roofline      dominant kernel = the decode weight-streaming GEMM (HBM-bound):
                algorithmic bytes = every LM + adapter + head weight byte once
                per token step (12.16 GB at full size) / time of the CAPTURED
                token step (the launches generate() replays: attention co-launches,
                argmax, bookkeeping included), measured live with HIP events on the
                launch stream; roofline.sweep_frac = the same weight streams as bare
                GEMVs back to back (the kernel family in isolation);
                roofline.train = the training half's dominant kernel (the 256x256
                MFMA GEMM at the four block-projection shapes, M = 16 x 2048),
                measured live the same way against 2.5 PF/s.
  cpu_baseline  the oracle (CPU restatement, "port") timed on the host cores on
                a bounded sample of the same workload, extrapolated linearly in
                layers (stated in "sample").
"""
import argparse
import datetime
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL / tensor sharing across processes on this driver); before the runtime loads
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--gen", type=int, default=32)
    ap.add_argument("--prompt", type=int, default=8)
    ap.add_argument("--res", type=int, default=224, help="image resolution (224 per BASELINE.json; 384 = model native)")
    ap.add_argument("--config", default="MAGMA_v1")
    ap.add_argument("--layers", type=int, default=None, help="debug: fewer LM layers (result is then NOT the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=5, help="timed training steps for the extra 'train' object (0 = skip)")
    ap.add_argument("--train-warmup", type=int, default=2, help="untimed training steps in front of the timed ones")
    ap.add_argument("--train-batch", type=int, default=16, help="per-GPU micro-batch of the training step (BASELINE config[2])")
    ap.add_argument("--train-truncate", action=argparse.BooleanOptionalAction, default=True,
                    help="also time the exact-truncation variant (SURVEY Q3: identical loss and gradients, the sequence is cut "
                         "after the longest caption); reported as train.truncated, never as the headline")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--fp8", choices=["attn", "all", "off"], default="all",
                    help="also time BASELINE config[4]: fp8 (e4m3) MFMA for QKV/out_proj/adapter GEMMs ('attn') or every "
                         "block GEMM ('all'), W8A16 decode and the fp8 training step; reported in extra objects "
                         "('generate_fp8', 'train.forward_only_fp8', 'train.full_S2048_fp8'), never in 'value'")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launch check: N ranks are started exactly as for a real run (self-launch or torch.distributed.run), join "
                         "the process group, run the bench's barrier / MAX-over-ranks collectives and rank 0 prints a line with "
                         "n_gpus = N and value = null; no model, no GPU needed with MAGMA_BENCH_BACKEND=gloo")
    ap.add_argument("--with-cpu-baseline", action="store_true",
                    help="--rendezvous-only: rank 0 also measures cpu_baseline where a real run does (after the last collective, the "
                         "other ranks waiting at the final barrier) -- the N > 1 control flow of that leg without a GPU")
    ap.add_argument("--train-only", action="store_true",
                    help="only the training leg (BASELINE's 'train images/sec, whole node'): metric / value / ms_per_step / roofline "
                         "of the line are the training step's; the generate legs, variants and config1 are skipped -- a multi-GPU "
                         "record that fits a short lease")
    ap.add_argument("--variants", action=argparse.BooleanOptionalAction, default=True,
                    help="N = 1 only: also time BASELINE config[3] (MAGMA_v2: attention + MLP adapters) and the model-native 384^2 "
                         "images as extra objects ('magma_v2', 'generate_res384'); never in 'value'")
    args = ap.parse_args()
    if args.fp8 == "off":
        args.fp8 = None
    if args.train_only:
        args.variants = False
        args.train_steps = max(1, args.train_steps)
    return args


def cpu_baseline(args, budget_s):
    """Oracle on the host cores ("port"): the CLIP trunk + prefix on one image (x batch), one full-size GPT-J
    block (+adapter) at the prefill shape and at the decode shape (x28 layers), plus lm_head -- timed in fp32 and
    in bf16, the faster one is `value`, both are reported.  Bounded to ~budget_s seconds."""
    from oracle import model as O
    cores = os.cpu_count()
    cfg = O.OracleConfig.magma_v1()
    cfg.n_layer = 1
    d = cfg.d_model
    B, P = args.batch, (args.res // 32) ** 2
    S0 = P + args.prompt
    h = "lm.transformer.h.0."
    a = h + "mlp.1.adapter."
    t_used = time.time()
    nthr = min(cores, 32)
    torch.set_num_threads(nthr)

    def build(dtype):
        g = torch.Generator().manual_seed(0)
        mk = lambda *s: (torch.randn(*s, generator=g) * 0.02).to(dtype)  # noqa: E731
        p = {h + "ln_1.weight": torch.ones(d, dtype=dtype), h + "ln_1.bias": torch.zeros(d, dtype=dtype)}
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            p[O.attn_prefix(cfg, 0) + n + ".weight"] = mk(d, d)
        mp = O.mlp_prefix(cfg, 0)
        p[mp + "c_fc.weight"], p[mp + "c_fc.bias"] = mk(cfg.d_ff, d), mk(cfg.d_ff)
        p[mp + "c_proj.weight"], p[mp + "c_proj.bias"] = mk(d, cfg.d_ff), mk(d)
        p[a + "0.weight"], p[a + "0.bias"] = mk(1024, d), mk(1024)
        p[a + "2.weight"], p[a + "2.bias"] = mk(d, 1024), mk(d)
        return p, mk

    L = 28
    legs = {}
    # thread counts: 32 (PyTorch's CPU GEMMs at these sizes -- 456 x 4096 x 16384 and 8 x 4096 x 16384 per call -- stop scaling
    # around one CCD group of the box's two sockets; more threads add NUMA traffic and barrier time) AND every hardware thread,
    # as far as the time budget goes: the faster (dtype, threads) leg is `value`, every leg is reported with its thread count
    plans = [(torch.float32, nthr), (torch.bfloat16, nthr)] + ([(torch.bfloat16, cores)] if cores > nthr else [])
    probe = None
    with torch.no_grad():
        # image encoder + prefix projection: the full RN50x16 trunk on ONE image in fp32, scaled by the batch
        enc_p = {k: v for k, v in O.init_params(O.OracleConfig(n_layer=0, vocab_in=8, vocab_out=8), seed=0).items()
                 if k.startswith("image_prefix.")}
        img = torch.randn(1, 3, args.res, args.res)
        ecfg = O.OracleConfig.magma_v1()
        O.image_prefix_fwd(enc_p, ecfg, img)
        t0 = time.time(); O.image_prefix_fwd(enc_p, ecfg, img); t_enc = (time.time() - t0) * B
        del enc_p
        for dtype, thr in plans:
            if legs and time.time() - t_used > budget_s * 0.6:
                break
            torch.set_num_threads(thr)
            p, mk = build(dtype)
            x1 = mk(B, 1, d) * 50
            if thr > nthr:
                # every hardware thread: ONE decode-shape block first.  On the 2-socket boxes of this pool it is two to three
                # orders of magnitude slower than at 32 threads (3.5 s against 3.7 ms per layer measured: the 8 x 4096 x 16384
                # GEMVs are split 256 ways across NUMA nodes and spend their time in barriers) -- then the leg stops here and
                # the probe is what gets reported, instead of spending half a minute on a number nobody would quote
                xs = mk(B, 4, d) * 50
                _, past_s = O.block_fwd(p, cfg, 0, xs, None, 0)
                t0 = time.time(); O.block_fwd(p, cfg, 0, x1, past_s, 4); tprobe = time.time() - t0
                ref = min(v["decode_layer_ms"] for v in legs.values())
                probe = {"threads": thr, "decode_layer_ms_one_call": tprobe * 1e3, "decode_layer_ms_at_%d_threads" % nthr: ref}
                del past_s
                if tprobe * 1e3 > 2.0 * ref:
                    del p
                    continue
            x = mk(B, S0, d) * 50
            O.block_fwd(p, cfg, 0, x, None, 0)
            t0 = time.time(); _, past = O.block_fwd(p, cfg, 0, x, None, 0); tp = time.time() - t0
            t0 = time.time()
            n_dec = 0
            while n_dec < 2 or (time.time() - t0 < budget_s * 0.12 and n_dec < 6):
                O.block_fwd(p, cfg, 0, x1, past, S0); n_dec += 1
            td = (time.time() - t0) / n_dec
            head_w, head_b = mk(cfg.vocab_out, d), mk(cfg.vocab_out)
            t0 = time.time(); torch.nn.functional.linear(x1[:, 0], head_w, head_b); th = time.time() - t0
            total = t_enc + L * tp + args.gen * (L * td + th)
            legs[f"{str(dtype).split('.')[-1]}@{thr}"] = {"tokens_per_s": B * args.gen / total, "threads": thr, "prefill_layer_ms": tp * 1e3,
                                                        "decode_layer_ms": td * 1e3, "lm_head_ms": th * 1e3}
            del p, head_w, head_b, past
    torch.set_num_threads(nthr)
    best = max(legs, key=lambda k: legs[k]["tokens_per_s"])
    return {"value": legs[best]["tokens_per_s"], "unit": "tokens/s", "cores": legs[best]["threads"], "kind": "port", "dtype": best.split("@")[0],
            "legs": legs, "all_threads_probe": probe, "encoder_ms_per_batch": t_enc * 1e3, "host_threads": cores,
            "sample": f"oracle (PyTorch CPU; legs at {nthr} threads, and at all {cores} hardware threads when a one-call probe there is not slower -- see all_threads_probe; the faster leg is value): CLIP RN50x16 trunk + "
                      f"prefix on 1 image x{B} ({t_enc*1e3:.0f} ms, fp32) + one full-size GPT-J block+adapter timed at prefill B={B},S={S0} "
                      f"and at the cached decode shape, x{L} layers + lm_head x {args.gen} steps (best leg: {best}); measured in {time.time()-t_used:.0f} s"}


def decode_gemv_jobs(eng, st):
    """Exactly the weight-streaming GEMV launches of one token step, on the real weights of every layer (12.16 GB, far beyond
    the 256 MiB Infinity Cache): (list of launch thunks, ALGORITHMIC weight bytes = every LM + adapter + head weight of the
    reference model once, [(N, K)] per launch as streamed).  With the folded block (engine.fold_dn) the launches stream
    [W_fc ; W_dn W_fc] and [W_out | W_up]: more bytes than the algorithm needs (reported as 'streamed_bytes'), never counted
    as achieved work."""
    from magma_amd import ops
    jobs, shapes = [], []
    algo = 0
    d3 = 3 * eng.d

    def add(fn, w):
        jobs.append(fn)
        shapes.append((w.N, w.K))

    for ly in eng.layers:
        add(lambda ly=ly: ops.gemm_skinny(st.xa, ly.dec_in, out=st.qkv, ln_fold=(ly.dec_in.colsum, eng.d, eng.eps),
                                          split=(d3, st.h, ops.MG_ACT_GELU_NEW, ly.dec_in.bias_b)), ly.dec_in)
        algo += 2 * (ly.dec_in.N * ly.dec_in.K + ly.out.N * ly.out.K + ly.fc_out.N * ly.fc_out.K)
        if ly.mlp_adapter:
            algo += 2 * sum(a.N * a.K for a in ly.mlp_adapter)
        if eng.fold_dn == 1 and getattr(ly, "fc_dn", None) is not None:
            r = ly.mlp_adapter[0].N
            t = st.ctx_t[:, eng.d: eng.d + r]
            add(lambda ly=ly, t=t: ops.gemm_skinny(st.h, ly.fc_dn, out=st.m, split=(eng.d, t, ops.MG_ACT_RELU, ly.fc_dn.bias_b)), ly.fc_dn)
            add(lambda ly=ly, r=r: ops.gemm_skinny(st.ctx_t[:, : eng.d + r], ly.out_up, out=st.xb, residuals=(st.m, st.xa)), ly.out_up)
            continue
        if eng.fold_dn == 2 and getattr(ly, "out_up", None) is not None:      # MAGMA_DECODE_FOLD=2: only the K-concatenated [W_out | W_up]
            r = ly.mlp_adapter[0].N
            add(lambda ly=ly: ops.gemm_skinny(st.h, ly.fc_out, out=st.m), ly.fc_out)
            add(lambda ly=ly, r=r: ops.gemm_skinny(st.m, ly.mlp_adapter[0], out=st.ctx_t[:, eng.d: eng.d + r], act=ops.MG_ACT_RELU), ly.mlp_adapter[0])
            add(lambda ly=ly, r=r: ops.gemm_skinny(st.ctx_t[:, : eng.d + r], ly.out_up, out=st.xb, residuals=(st.m, st.xa)), ly.out_up)
            continue
        add(lambda ly=ly: ops.gemm_skinny(st.ctx, ly.out, out=st.a), ly.out)
        add(lambda ly=ly: ops.gemm_skinny(st.h, ly.fc_out, out=st.m), ly.fc_out)
        if ly.mlp_adapter:
            r = ly.mlp_adapter[0].N
            add(lambda ly=ly, r=r: ops.gemm_skinny(st.m, ly.mlp_adapter[0], out=st.t[:, :r], act=ops.MG_ACT_RELU), ly.mlp_adapter[0])
            add(lambda ly=ly, r=r: ops.gemm_skinny(st.t[:, :r], ly.mlp_adapter[1], out=st.xb, residuals=(st.m, st.a, st.xa)), ly.mlp_adapter[1])
    add(lambda: ops.gemm_skinny(st.xa, eng.head_dec, out=st.logits, ln_fold=(eng.head_dec.colsum, eng.d, eng.eps)), eng.head_dec)
    algo += 2 * eng.head_dec.N * eng.head_dec.K
    return jobs, algo, shapes


def pmc_traffic_per_launch(shapes):
    """HBM bytes per launch of the GEMV sweep from the committed rocprofv3 FETCH_SIZE pass over this very sweep
    (profiles/rNN_decode_gemv_fetch_table.json, newest round; tools/gpu_pmc_decode.sh + tools/pmc_decode_sweep.py: per (N, K)
    shape, counter x 2 for the gfx950 wide-read under-count, MI355X_MICROARCH.md HBM).  None when a shape is missing."""
    import glob
    tabs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_decode_gemv_fetch_table.json")))
    if not tabs:
        return None, None
    path = tabs[-1]
    tab = json.load(open(path))["bytes_per_launch"]
    try:
        tot = sum(tab[f"{n}x{k}"] for n, k in shapes)
    except KeyError:
        return None, None
    return tot / len(shapes), os.path.relpath(path, ROOT)


def train_flops_per_image(model, res, S, c=1.0, bottom_rows=0):
    """SURVEY 8d, 'no recompute' policy: T = 2G + 3A + W + 3E per image.  c = 1: full-S^2 attention FLOPs, as the
    reference computes them; c = 0.5: the causal tiles the flash kernels actually execute.  ``bottom_rows`` = P > 0 (executed
    count only): the bottom block of the frozen LM forms its input gradient for the P prefix rows only -- the dgrads through
    qkv / fc_in / fc_out of that block run on P of S rows, its dQ on the first ceil(P / 128) query blocks and its dK / dV on the
    first key blocks (train_engine._lm_backward)."""
    L, d, ff, V = model.lm.config.num_layers, model.lm.config.hidden_size, model.lm.config.intermediate_size, model.lm.config.vocab_size
    r = sum(ad.N for ad in (model.lm.engine.layers[0].mlp_adapter or ())[:1]) + sum(ad.N for ad in (model.lm.engine.layers[0].attn_adapter or ())[:1])
    G = S * (L * (8 * d * d + 4 * d * ff + 4 * d * r))            # block GEMMs fwd (head runs on target rows only)
    A = 4 * L * d * S * S * c                                      # attention fwd
    Wg = 4 * S * L * d * r                                         # adapter wgrad
    E = 47.72e9 * (res / 224.0) ** 2                               # CLIP trunk fwd per image
    skipped = 0.0
    if bottom_rows:
        nb, n = (S + 127) // 128, (bottom_rows + 127) // 128
        f = n / nb
        skipped = (S - bottom_rows) * (6 * d * d + 4 * d * ff)     # three dgrad GEMMs of block 0 on the rows that are not formed
        # attention backward of block 0 (2 x its forward): dK / dV of the first n key blocks (every later query: f (2 - f) of the
        # causal tiles, ~0.6 of the backward's products), dQ of the first n query blocks (f^2, ~0.4)
        skipped += 2 * (4 * d * S * S * c) * (1.0 - (0.6 * f * (2 - f) + 0.4 * f * f))
    return 2 * G + 3 * A + Wg + 3 * E - skipped


def _forward_fp8(model, images, caps, mode, sync, dtf, f_fwd, scaling="row"):
    """BASELINE config[4]: the same forward with the fp8 projections (inference engine; the backward stays bf16).
    scaling "mx": OCP MX block scales (one E8M0 per 32 K-elements of both operands) instead of per-row / per-channel fp32 scales."""
    lm_eng = model.lm.engine
    old_scaling = lm_eng.fp8_scaling
    lm_eng.fp8_scaling = scaling
    try:
        with torch.no_grad():
            ref_loss = float(model(images, caps).loss)
            lm_eng.fp8_mode = mode
            l8 = float(model(images, caps).loss)      # packs the e4m3 weights
            sync()
            t0 = time.perf_counter()
            for _ in range(2):
                model(images, caps)
            sync()
            dt8 = (time.perf_counter() - t0) / 2
        if scaling == "mx":
            return {"mode": mode, "scaling": "mx", "ms": dt8 * 1e3, "speedup_vs_bf16": dtf / dt8, "loss_bf16": ref_loss, "loss_fp8": l8,
                    "algorithmic_tflops": f_fwd / dt8 / 1e12,
                    "note": "the same forward with OCP MX block scales (E8M0 per 32 K-elements of activations and weights, applied "
                            "by the MFMA; 256x256 kernel with the scales staged through LDS, round 5)"}
        return {"mode": mode, "ms": dt8 * 1e3, "speedup_vs_bf16": dtf / dt8, "loss_bf16": ref_loss, "loss_fp8": l8,
                "algorithmic_tflops": f_fwd / dt8 / 1e12,
                "note": "e4m3 operands with per-row / per-channel fp32 scales on v_mfma_scale_f32_16x16x128_f8f6f4 "
                        "(unit block scales), fp32 accumulate, bf16 I/O; attention forward on v_mfma_scale_f32_32x32x64_f8f6f4 "
                        "(MX e4m3 q / k / v^T, P as e4m3, fp32 softmax) unless MAGMA_FP8_ATTN=0"}
    except Exception as e:  # noqa: BLE001  (the bf16 numbers must survive a failure of the extra leg)
        return {"error": repr(e)[:300]}
    finally:
        lm_eng.fp8_mode = None
        lm_eng.fp8_scaling = old_scaling


def timed_steps(step, n, warmup, sync):
    """`warmup` untimed steps, then n steps bracketed by sync() (the mean, as before) with one HIP event between steps on the
    launch stream: (mean seconds, {"min_ms", "median_ms", "max_ms"}, last return value)."""
    for _ in range(warmup):
        step()
    sync()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        ret = step()
        ev[i + 1].record()
    sync()
    dt = (time.perf_counter() - t0) / n
    per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return dt, {"min_ms": per[0], "median_ms": per[len(per) // 2] if n % 2 else 0.5 * (per[n // 2 - 1] + per[n // 2]),
                "max_ms": per[-1], "timed_steps": n, "warmup_steps": warmup}, ret


def bench_train(model, args, rank, world, dev):
    """Config[2]: MAGMA_v1 training step, synthetic img-caption pairs, per-GPU batch 16,
    S = 2048, trainable = adapters + CLIP trunk + prefix; no recompute; AdamW + clip inside
    the timed region; DP all-reduce when world > 1."""
    from magma_amd.datasets import synthetic_batch
    from magma_amd.train_engine import MagmaEngine
    model.config.gradient_accumulation_steps = 1
    out = {}
    eng = MagmaEngine(model)
    eng.train()
    B, S = args.train_batch, model.seq_len
    images, caps = synthetic_batch(B, args.res, S, model.eos_token, 50256, 1234 + rank, device=dev, dtype=torch.bfloat16)
    caps_host = caps.cpu()      # what a data loader hands over: the label index plumbing is done on the host (no sync)

    def step():
        o = eng(images, caps, captions_host=caps_host)
        eng.backward(o.loss)
        eng.step()
        return o.loss

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # forward only (eval mode, same batch): the north-star's "GPT-J + adapter forward" MFMA fraction
    eng.eval()
    with torch.no_grad():
        model(images, caps)
        sync()
        t0 = time.perf_counter()
        for _ in range(2):
            model(images, caps)
        sync()
    dtf = (time.perf_counter() - t0) / 2
    L, d, ff = model.lm.config.num_layers, model.lm.config.hidden_size, model.lm.config.intermediate_size
    r = sum(ad.N for ad in (model.lm.engine.layers[0].mlp_adapter or ())[:1]) + sum(ad.N for ad in (model.lm.engine.layers[0].attn_adapter or ())[:1])
    f_fwd = B * (S * L * (8 * d * d + 4 * d * ff + 4 * d * r) + 4 * L * d * S * S + 47.72e9 * (args.res / 224.0) ** 2)
    f_fwd_exec = f_fwd - B * 2 * L * d * S * S          # causal tiles only (c = 1/2): what the kernels execute
    out["forward_only"] = {"ms": dtf * 1e3, "algorithmic_tflops": f_fwd / dtf / 1e12, "mfma_frac_of_2.5PF": f_fwd / dtf / 2.5e15,
                           "executed_tflops": f_fwd_exec / dtf / 1e12, "mfma_frac_executed": f_fwd_exec / dtf / 2.5e15,
                           "note": "image prefix + 28 blocks at S=2048 + loss on target rows; 'algorithmic' counts full-S^2 attention "
                                   "FLOPs as the reference computes them (c = 1), 'executed' the causal tiles the kernel runs (c = 1/2)"}
    if args.fp8:
        out["forward_only_fp8"] = _forward_fp8(model, images, caps, args.fp8, sync, dtf, f_fwd)
        out["forward_only_fp8_mx"] = _forward_fp8(model, images, caps, args.fp8, sync, dtf, f_fwd, scaling="mx")
    eng.train()
    exposed_comm_ms = overlapped = comm_busy_ms = train_per_rank = None
    for trunc in ([False, True] if args.train_truncate else [False]):
        eng.truncate = trunc
        for _ in range(args.train_warmup):
            step()
        sync()
        eng.time_comm = not trunc
        eng.exposed_comm_ms()
        dt, spread, loss = timed_steps(step, args.train_steps, 0, sync)
        if not trunc:
            exposed_comm_ms, overlapped = eng.exposed_comm_ms(), getattr(eng, "last_overlapped_elems", None)
        if not trunc:
            comm_busy_ms = eng.comm_busy_ms() if hasattr(eng, "comm_busy_ms") else None
        eng.time_comm = False
        if world > 1:
            if not trunc:
                train_per_rank = per_rank_ms(dt, dev)
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t)
        key = "truncated" if trunc else "full_S2048"
        fl = train_flops_per_image(model, args.res, S) * B
        fl_exec = train_flops_per_image(model, args.res, S, c=0.5, bottom_rows=getattr(eng, "bottom_prefix_rows", 0)) * B
        out[key] = {"images_per_s": world * B / dt, "ms_per_step": dt * 1e3, "loss": float(loss),
                    "algorithmic_tflops_per_gpu": None if trunc else fl / dt / 1e12,
                    "mfma_frac_of_2.5PF": None if trunc else fl / dt / 2.5e15,
                    "mfma_frac_executed": None if trunc else fl_exec / dt / 2.5e15, "spread": spread}
        if trunc:
            out[key]["note"] = ("NOT the BASELINE config: the sequence is cut after the longest caption (+ prefix); with causal "
                                "attention and the masked loss this leaves loss and gradients mathematically unchanged (same values up "
                                "to bf16 summation order: tests/test_train_gpu.py::test_truncation_is_exact, loss within 2e-3, gradients "
                                "within 2e-2 rel-L2) but skips the padded positions the reference computes")
    eng.truncate = False
    if args.fp8:   # BASELINE config[4], training side: frozen-weight block GEMMs (forward + dgrad) on the fp8 MFMA
        eng.fp8, eng.truncate = True, False
        try:
            dt8, spread8, loss8 = timed_steps(step, args.train_steps, args.train_warmup, sync)
            if world > 1:
                t = torch.tensor([dt8], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                dt8 = float(t)
            out["full_S2048_fp8"] = {"images_per_s": world * B / dt8, "ms_per_step": dt8 * 1e3, "loss": float(loss8), "spread": spread8,
                                     "note": "qkv / out_proj / fc_in / fc_out forward and dgrad GEMMs in e4m3 (per-row / per-channel "
                                             "scales, fp32 accumulate) and the attention forward (QK^T / PV on the MX-scaled fp8 MFMA, OCP e4m3 operands); "
                                             "adapters, attention backward, wgrads, trunk stay bf16",
                                     "loss_note": "this leg runs AFTER the optimizer steps of the bf16 legs (same model, same batch), so its "
                                                  "loss is further down the training curve, not comparable with full_S2048.loss; the same-weights "
                                                  "comparison is train.forward_only_fp8 (loss_bf16 / loss_fp8)"}
        except Exception as e:  # noqa: BLE001
            out["full_S2048_fp8"] = {"error": repr(e)[:300]}
        finally:
            eng.fp8 = False
    out["policy"] = ("no recompute; bf16; adapters+CLIP trunk+prefix trainable; clip 1.0 + AdamW in the timed region; the bottom LM block forms its "
                     "input gradient for the image-prefix rows only (nothing else below it is trainable: every parameter gradient unchanged)")
    out["per_gpu_batch"], out["seq_len"] = B, S
    n_train = sum(g.n for g in eng.groups)
    backend = None
    if world > 1:
        backend = getattr(eng._exchange, "name", "?") + " / " + torch.distributed.get_backend() + (" = RCCL" if torch.distributed.get_backend() == "nccl" else "")
    out["data_parallel"] = {"ranks": world, "backend": backend,
                            "rccl_ranks": torch.distributed.get_world_size() if world > 1 else None,
                            "gradient_exchange": ("bf16 buckets" if eng.exchange_bf16 else "fp32") if world > 1 else None,
                            "exchanged_elements": n_train if world > 1 else None,
                            "elements_handed_over_during_backward": overlapped if world > 1 else None,
                            "exposed_comm_ms_per_step": exposed_comm_ms,
                            # the exchange stream's own busy time per step (HIP events around every bucket's all-reduce on it): with
                            # exposed ~ 0 and busy >> the 9 ms a 0.77 GB ring needs, the buckets are waiting for CUs behind the GEMMs
                            # (MAGMA_DP_RESERVE_CUS leaves some out of the compute stream's mask)
                            "comm_stream_busy_ms_per_step": comm_busy_ms,
                            "per_rank_step_ms": train_per_rank,      # this rank-0 line carries every rank's own step time
                            "reserved_cus": int(os.environ.get("MAGMA_DP_RESERVE_CUS", "0")),
                            "global_batch": world * B}
    out["max_memory_allocated_GB"] = torch.cuda.max_memory_allocated() / 2 ** 30
    return out


def gemm_roofline(model, args, dev):
    """The dominant kernel of the TRAINING half of the metric, live: the 256x256 MFMA GEMM on the four frozen-weight projections
    of a GPT-J block at the training shape (M = per-GPU batch x 2048 rows; qkv, out_proj, fc_in + gelu_new, fc_out), random
    operands, layer-0 weights, HIP events on the launch stream.  achieved = 2 M N K summed / time."""
    from magma_amd import ops
    eng = model.lm.engine
    ly = eng.layers[0]
    M = args.train_batch * model.seq_len
    g = torch.Generator(device=dev).manual_seed(7)
    a_d = torch.randn(M, eng.d, device=dev, generator=g).to(torch.bfloat16)
    a_ff = torch.randn(M, ly.fc_in.N, device=dev, generator=g).to(torch.bfloat16)
    jobs = [(a_d, ly.qkv, {}), (a_d, ly.out, {}), (a_d, ly.fc_in, {"act": ops.MG_ACT_GELU_NEW}), (a_ff, ly.fc_out, {})]
    outs = [torch.empty(M, w.N, dtype=torch.bfloat16, device=dev) for _, w, _ in jobs]
    flops = sum(2.0 * M * w.N * w.K for _, w, _ in jobs)

    def sweep():
        for (a, w, kw), o in zip(jobs, outs):
            ops.gemm(a, w, out=o, **kw)
    sweep()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        sweep()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = flops / ms / 1e9
    out = {"bound": "mfma", "kernel": "gemm256_kernel (256x256x64 tiles, v_mfma_f32_16x16x32_bf16)", "achieved": tf, "peak": 2500.0,
           "unit": "TFLOP/s", "frac": tf / 2500.0, "launches": len(jobs), "avg_launch_us": ms * 1e3 / len(jobs),
           "shapes_MxNxK": [[M, w.N, w.K] for _, w, _ in jobs]}
    out["sustained_clock"] = sustained_clock_note(tf)
    return out


def sustained_clock_note(tf):
    """The 2.5 PF/s peak is 1024 FLOP/clk/SIMD at 2.4 GHz; under the tile GEMM's own load the chip clocks to its power budget.
    The clock is measured INSIDE the kernel (s_memtime cycles of a workgroup's life / its 100-MHz wall time; ablation build of
    the library, tools/kbench.py stamps) and committed as a profile -- not measured in this run.  Returns the time-weighted
    clock over the four projection shapes, the matrix peak at that clock and this run's achieved rate against it."""
    import glob
    tabs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm256_shader_clock.jsonl")))
    if not tabs:
        return None
    rows = {}
    for ln in open(tabs[-1]):
        try:
            d = json.loads(ln)
        except ValueError:
            continue
        if d.get("kind") == "stamps" and d.get("tag") in ("qkv", "out_proj", "fc_in", "fc_out"):
            rows.setdefault(d["tag"], []).append((d["shader_clock_MHz"]["mean"], d["launch_span_us"]))
    if len(rows) < 4:
        return None
    wsum = sum(t for v in rows.values() for _, t in v)
    clk = sum(c * t for v in rows.values() for c, t in v) / wsum
    peak = 2500.0 * clk / 2400.0
    return {"MHz_time_weighted": round(clk, 1), "MHz_by_shape": {k: round(sum(c for c, _ in v) / len(v), 1) for k, v in rows.items()},
            "peak_at_this_clock_TFLOPs": round(peak, 1), "frac_of_peak_at_this_clock": round(tf / peak, 3),
            "source": os.path.relpath(tabs[-1], ROOT),
            "note": "clock measured in-kernel on the instrumented (ablation) build of the same kernel, same shapes, random operands; "
                    "the kernel of this run draws at least that power, so its own clock is no higher"}


def gemm_roofline_fp8(model, args, dev):
    """BASELINE config[4]'s GEMM against ITS peak, live: the same four projections on the fp8 form of the 256x256 kernel (OCP e4m3 operands,
    per-row activation scales from mg_quantize_rows_fp8 -- outside the timed region, as in the training step where one quantised copy
    feeds qkv and fc_in -- per-output-channel weight scales, fp32 accumulate, bf16 output) against the 5 PF/s dense fp8 MFMA peak."""
    from magma_amd import ops
    eng = model.lm.engine
    ly = eng.layers[0]
    M = args.train_batch * model.seq_len
    g = torch.Generator(device=dev).manual_seed(7)
    a_d = torch.randn(M, eng.d, device=dev, generator=g).to(torch.bfloat16)
    a_ff = torch.randn(M, ly.fc_in.N, device=dev, generator=g).to(torch.bfloat16)
    q_d, q_ff = ops.quantize_rows_fp8(a_d), ops.quantize_rows_fp8(a_ff)
    del a_d, a_ff

    def w8(lin):
        return ops.PackedLinearFP8(ops.PackedLinear.untile(lin.ft)[: lin.N, : lin.K], lin.bias)
    jobs = [(q_d, w8(ly.qkv), {}), (q_d, w8(ly.out), {}), (q_d, w8(ly.fc_in), {"act": ops.MG_ACT_GELU_NEW}), (q_ff, w8(ly.fc_out), {})]
    outs = [torch.empty(M, w.N, dtype=torch.bfloat16, device=dev) for _, w, _ in jobs]
    flops = sum(2.0 * M * w.N * w.K for _, w, _ in jobs)

    def sweep():
        for ((q, sc), w, kw), o in zip(jobs, outs):
            ops.gemm_fp8(q, sc, w, out=o, **kw)
    sweep()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        sweep()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = flops / ms / 1e9
    return {"bound": "mfma", "kernel": "gemm256_kernel<FP8> (256x256 tiles, v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales)", "achieved": tf,
            "peak": 5000.0, "unit": "TFLOP/s", "frac": tf / 5000.0, "launches": len(jobs), "avg_launch_us": ms * 1e3 / len(jobs),
            "shapes_MxNxK": [[M, w.N, w.K] for _, w, _ in jobs]}


def variant_generate(model, args, dev, res):
    """The headline call at another image resolution (384^2 = the model's own: 144 prefix tokens, prefill S0 = 152)."""
    B, gen = args.batch, args.gen
    g = torch.Generator(device=dev).manual_seed(4321)
    images = torch.randn(B, 3, res, res, device=dev, generator=g).to(torch.bfloat16)
    prompt = torch.randint(0, 50256, (B, args.prompt), device=dev, generator=g)

    def call():
        return model.generate(model.embed([images, prompt]), max_steps=gen, temperature=0.0, decode=False, stop_on_eos=False)
    toks = call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    return {"tokens_per_s": B * gen / dt, "ms_per_call": dt * 1e3, "prefill_len": int(toks.shape[1] - gen), "resolution": res}


def variant_v2(args, dev):
    """BASELINE config[3]: MAGMA_v2.yml (attention AND MLP adapters, downsample 8) -- the same generate call and training step."""
    import gc
    from magma_amd import Magma
    from magma_amd.datasets import synthetic_batch
    from magma_amd.train_engine import MagmaEngine
    gc.collect()
    torch.cuda.empty_cache()
    torch.manual_seed(1234)
    model = Magma("MAGMA_v2", device=dev)
    model.eval()
    out = {"generate": variant_generate(model, args, dev, args.res)}
    if args.train_steps > 0:
        model.config.gradient_accumulation_steps = 1
        eng = MagmaEngine(model)
        eng.train()
        B, S = args.train_batch, model.seq_len
        images, caps = synthetic_batch(B, args.res, S, model.eos_token, 50256, 1234, device=dev, dtype=torch.bfloat16)
        caps_host = caps.cpu()

        def step():
            o = eng(images, caps, captions_host=caps_host)
            eng.backward(o.loss)
            eng.step()
            return o.loss
        dt, spread, loss = timed_steps(step, args.train_steps, args.train_warmup, torch.cuda.synchronize)
        fl = train_flops_per_image(model, args.res, S) * B
        out["train_full_S2048"] = {"images_per_s": B / dt, "ms_per_step": dt * 1e3, "loss": float(loss), "spread": spread,
                                   "mfma_frac_of_2.5PF": fl / dt / 2.5e15,
                                   "mfma_frac_executed": train_flops_per_image(model, args.res, S, c=0.5, bottom_rows=getattr(eng, "bottom_prefix_rows", 0)) * B / dt / 2.5e15}
    return out


def flat_training_keys(train, world, args):
    """What a scaling curve over N is computed from, at the TOP level of the line (a flat parser keeps these): whole-job training
    throughput, every rank's own step time, the gradient exchange's exposed time and the exchange stream's busy time per step
    (reference train.py:103-111, magma/utils.py:26-34: the step is one DP all-reduce wide)."""
    full = (train or {}).get("full_S2048") or {}
    dpo = (train or {}).get("data_parallel") or {}
    return {"train_images_per_s": full.get("images_per_s"), "train_ms_per_step": full.get("ms_per_step"), "train_ranks": world,
            "train_per_gpu_batch": args.train_batch,
            "train_per_rank_step_ms": dpo.get("per_rank_step_ms") or ([round(full["ms_per_step"], 4)] if full.get("ms_per_step") else None),
            "train_exposed_comm_ms": dpo.get("exposed_comm_ms_per_step"),
            "train_comm_stream_busy_ms": dpo.get("comm_stream_busy_ms_per_step")}


def config1_leg(model, dev):
    """BASELINE config[0] at its stated shape on the model of this run (reference README.md:84, magma/magma.py:176-212): one
    224 x 224 image FILE + an 8-token prompt -> preprocess_inputs (resize to the model-native resolution, prefix tokens) ->
    (1, P + 8, 4096) -> LM forward.  Pass = shapes as stated + finite logits + the caller's list mutated as the reference does;
    wall seconds of the second call (the first pays one-time packing).  The reference runs this on device='cpu'; parity of the
    same path against the oracle is tests/test_config1_gpu.py."""
    import tempfile
    import numpy as np
    import PIL.Image as I
    from magma_amd import ImageInput
    model.eval()                 # the training leg ran before this one: back to the inference operands (BatchNorm folded, packed weights)
    rng = np.random.RandomState(3)
    arr = (rng.rand(224, 224, 3) * 255).astype(np.uint8)
    prompt = "Describe"          # 8 tokens under the byte-level stand-in tokenizer (no GPT-2 files offline); preprocess_inputs takes str / ImageInput only
    n_tok = int(model.tokenizer.encode(prompt, return_tensors="pt").shape[1])
    P = model.image_prefix_seq_len
    d = model.lm.config.hidden_size
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "image.png")
        I.fromarray(arr).save(path)
        wall = None
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            inputs = [ImageInput(path), prompt]
            with torch.no_grad():
                emb = model.preprocess_inputs(inputs)
                logits = model.lm(inputs_embeds=emb).logits
            finite = bool(torch.isfinite(logits.float()).all())
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
    res = model.image_prefix.enc.input_resolution
    ok = (tuple(emb.shape) == (1, P + n_tok, d) and tuple(logits.shape)[:2] == (1, P + n_tok) and finite
          and tuple(inputs[0].shape) == (1, 3, res, res) and tuple(inputs[1].shape) == (1, n_tok))
    return {"pass": bool(ok), "wall_s": wall, "embeddings_shape": list(emb.shape), "logits_shape": list(logits.shape),
            "layers": model.lm.config.num_layers, "image": "224x224 RGB file", "prompt_tokens": n_tok, "device": str(dev),
            "note": "BASELINE config[0] (plumbing) on the GPU: this build has no CPU execution path; parity vs the oracle in tests/test_config1_gpu.py"}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, the way the reference is
    started (one process per GPU: reference README.md:121 `deepspeed train.py`, train.py:76,103-111) -- re-exec this script
    under torch.distributed.run on 127.0.0.1 with the same flags.  Returns the launcher's exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.call(cmd, env=env)


def check_launch(args, world):
    """A mis-launched run must not pass for an N-GPU run: --gpus has to equal the number of ranks, and (outside the one-GPU
    rehearsal mode, MAGMA_BENCH_DEVICE) every rank needs its own visible GPU."""
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                 f"(or run `python bench.py --gpus {args.gpus}` without a launcher: it starts the ranks itself)")
    if args.rendezvous_only or "MAGMA_BENCH_DEVICE" in os.environ:
        return
    n_dev = torch.cuda.device_count()
    if n_dev < world:
        sys.exit(f"bench.py: --gpus {world} but only {n_dev} GPU(s) visible; refusing to label a {n_dev}-GPU run n_gpus={world}")


def per_rank_ms(dt_local, dev, div=1):
    """Every rank's own wall time of the timed region (ms, per step), gathered to all ranks: the line then says which rank
    was the slow one (the headline uses the MAX) -- the first thing to look at when the N-GPU number disappoints."""
    import torch.distributed as dist
    t = torch.tensor([dt_local * 1e3 / div], dtype=torch.float64, device=dev if dev is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [round(float(x), 4) for x in out]


def rendezvous_only(args, rank, world):
    """The multi-rank control flow of main() without the model: process group, barrier, MAX over ranks, rank-0 line."""
    import torch.distributed as dist
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank = [dt * 1e3]
    if world > 1:
        per_rank = per_rank_ms(dt, None)
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        ranks = torch.zeros(world, dtype=torch.int64)
        ranks[rank] = 1
        dist.all_reduce(ranks)
        assert int(ranks.sum()) == world
    if rank == 0:
        line = {"metric": "launch check (no model)", "value": None, "unit": "tokens/s", "n_gpus": world, "steps": 0,
                "warmup": 0, "ms_per_step": dt * 1e3, "per_rank_ms": per_rank, "rendezvous_only": True,
                "backend": dist.get_backend() if world > 1 else None,
                "launched_by": os.environ.get("MAGMA_BENCH_LAUNCHER", "external")}
        line.update(flat_training_keys(None, world, args))
        if args.with_cpu_baseline:
            # as in main(): rank 0 alone, after the last collective of the timed part, the other ranks parked at the final barrier
            line["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        os.environ["MAGMA_BENCH_LAUNCHER"] = "self"
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    check_launch(args, world)
    if args.rendezvous_only:
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(os.environ.get("MAGMA_BENCH_BACKEND", "nccl"), rank=rank, world_size=world,
                                    timeout=datetime.timedelta(minutes=30))    # rank 0 measures the roofline objects alone while the others wait
        return rendezvous_only(args, rank, world)
    # one rank per GPU over RCCL.  MAGMA_BENCH_DEVICE / MAGMA_BENCH_BACKEND exist for ONE purpose: rehearsing the multi-rank
    # control flow (collective order, barriers, rank-0-only sections) with two processes on a single-GPU box (gloo, both
    # ranks on device 0) -- tools/gpu_bench_2rank_rehearsal.sh; numbers from such a run mean nothing
    local = int(os.environ.get("MAGMA_BENCH_DEVICE", os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("MAGMA_BENCH_BACKEND", "nccl"), rank=rank, world_size=world,
                                    timeout=datetime.timedelta(minutes=30))    # rank 0 measures the roofline objects alone while the others wait
    from magma_amd import Magma
    from magma_amd.language_model import GPTJConfig

    torch.manual_seed(1234)            # same random-init weights on every rank (the data below is rank-dependent)
    lm_cfg = None
    if args.layers is not None:
        lm_cfg = GPTJConfig(num_layers=args.layers, vocab_size=50258)
    model = Magma(args.config, device=dev, lm_config=lm_cfg)
    model.eval()
    eng = model.lm.engine
    B, gen = args.batch, args.gen
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn(B, 3, args.res, args.res, device=dev, generator=g).to(torch.bfloat16)
    prompt = torch.randint(0, 50256, (B, args.prompt), device=dev, generator=g)

    value = ms_step = roof = gen_s = gen_h = gen8 = gen_per_rank = toks = None
    if not args.train_only:
        def one_step():
            emb = model.embed([images, prompt])
            return model.generate(emb, max_steps=gen, temperature=0.0, decode=False, stop_on_eos=False)

        def sync():
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            one_step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            toks = one_step()
        sync()
        dt = time.perf_counter() - t0
        gen_per_rank = [round(dt / args.steps * 1e3, 4)]
        if world > 1:
            gen_per_rank = per_rank_ms(dt, dev, args.steps)
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t)
        ms_step = dt / args.steps * 1e3
        value = world * B * gen / (dt / args.steps)

        # ---- roofline of the dominant kernel (decode weight streaming), measured live ----
        roof = None
        if rank == 0:
            emb = model.embed([images, prompt])
            out = model.lm(inputs_embeds=emb, use_cache=True, cache_hint=gen + 40)
            cache = out.past_key_values
            tok = out.logits[:, -1].argmax(-1, keepdim=True)
            for _ in range(3):                                       # eager, capture, first replay
                eng.decode(tok, cache)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_rep = 20
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n_rep):
                eng.decode(tok, cache)
            e1.record()
            torch.cuda.synchronize()
            ms_tok = e0.elapsed_time(e1) / n_rep
            # dominant kernel in isolation: every decode GEMV of the model (all layers + head) launched
            # back to back on the real weights (12.16 GB, far beyond the 256 MiB Infinity Cache), HIP events
            # on the launch stream.  achieved = algorithmic bytes per launch / average launch duration.
            st = cache.decode_state
            jobs, wbytes, shapes = decode_gemv_jobs(eng, st)

            def sweep():
                for fn in jobs:
                    fn()

            gk = torch.cuda.CUDAGraph()
            sweep()
            torch.cuda.synchronize()
            with torch.cuda.graph(gk):
                sweep()
            gk.replay()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                gk.replay()
            e1.record()
            torch.cuda.synchronize()
            ms_sweep = e0.elapsed_time(e1) / 10
            sweep_achieved = wbytes / (ms_sweep * 1e-3) / 1e9
            achieved = wbytes / (ms_tok * 1e-3) / 1e9      # the launches the captured token step RUNS (attention co-launches, argmax, bookkeeping included)
            traffic, traffic_src = pmc_traffic_per_launch(shapes)
            roof = {"bound": "hbm", "kernel": "skinny_kernel (decode weight-streaming GEMM, M=8): the captured token step",
                    "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                    # the same weight streams as bare GEMVs back to back (fc_out as a plain GEMV instead of the attention co-launch)
                    "sweep_achieved": sweep_achieved, "sweep_frac": sweep_achieved / 8000.0,
                    # HBM bytes per launch from the PMC pass over this sweep (per GEMV shape), not measured in this run
                    "traffic": traffic, "traffic_source": traffic_src,
                    "bytes_per_launch": wbytes / len(jobs), "launches": len(jobs),
                    "streamed_bytes": sum(n * k * 2 for n, k in shapes), "algorithmic_bytes": wbytes,
                    "block": {0: "4 launches", 1: "3 launches (adapter-down folded through fc_out)", 2: "4 launches, [W_out | W_up] K-concatenated"}.get(eng.fold_dn, "?"),
                    "avg_launch_us": ms_sweep * 1e3 / len(jobs),
                    "token_step": {"ms": ms_tok, "decode_tokens_per_s": B / (ms_tok * 1e-3),
                                   "hbm_frac_whole_step": wbytes / (ms_tok * 1e-3) / 8e12}}

        # the reference's DEFAULT generate() settings (temperature 0.7, top_p 0.9, magma.py:214-221): the sampled branch -- top-p
        # rule, softmax, multinomial -- runs inside the same captured token step (csrc/sampling.hip); never the headline
        gen_s = None
        try:
            def sampled_step():
                emb = model.embed([images, prompt])
                return model.generate(emb, max_steps=gen, temperature=0.7, top_k=0, top_p=0.9, decode=False, stop_on_eos=False, seed=1)
            sampled_step()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                sampled_step()
            sync()
            dts = (time.perf_counter() - t0) / args.steps
            gen_s = {"mode": "temperature 0.7, top_p 0.9 (reference defaults), device-side sampling in the decode graph",
                     "tokens_per_s": world * B * gen / dts, "ms_per_call": dts * 1e3, "vs_greedy": (dt / args.steps) / dts}
        except Exception as e:  # noqa: BLE001
            gen_s = {"error": repr(e)[:300]}

        # the reference's callers hand over HOST tensors (preprocess_inputs -> CPU float images); `value` above starts with the
        # inputs resident in HBM, this leg adds the H2D copy + cast of the batch (pinned fp32 images, int64 prompt) to every call
        gen_h = None
        try:
            images_h, prompt_h = images.float().cpu().pin_memory(), prompt.cpu().pin_memory()

            def host_step():
                emb = model.embed([images_h.to(dev, non_blocking=True).to(torch.bfloat16), prompt_h.to(dev, non_blocking=True)])
                return model.generate(emb, max_steps=gen, temperature=0.0, decode=False, stop_on_eos=False)
            host_step()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                host_step()
            sync()
            dth = (time.perf_counter() - t0) / args.steps
            gen_h = {"mode": "inputs start in pinned host memory (fp32 images, int64 prompt): PCIe copy + cast inside the timed region",
                     "tokens_per_s": world * B * gen / dth, "ms_per_call": dth * 1e3, "h2d_bytes_per_call": int(images_h.numel() * 4 + prompt_h.numel() * 8)}
        except Exception as e:  # noqa: BLE001
            gen_h = {"error": repr(e)[:300]}

        gen8 = None
        if args.fp8:
            # BASELINE config[4] on the inference side: e4m3 weights in every decode GEMV (W8A16: bf16 activations, weights
            # widened in registers -> half the bytes per token step) + fp8 MFMA projections in the prefill.  Different
            # numerics (weight quantisation), so this is a separate object and never the headline `value`.
            eng.decode_w8, eng.fp8_mode = True, args.fp8
            eng._cache_pool.clear()
            try:
                one_step()
                sync()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    one_step()
                sync()
                dt8 = (time.perf_counter() - t0) / args.steps
                gen8 = {"mode": f"decode W8A16 + prefill fp8 '{args.fp8}'", "tokens_per_s": world * B * gen / dt8, "ms_per_call": dt8 * 1e3,
                        "speedup_vs_bf16": (dt / args.steps) / dt8}
            except Exception as e:  # noqa: BLE001  (the bf16 line must survive a failure of the extra leg)
                gen8 = {"error": repr(e)[:300]}
            finally:
                eng.decode_w8, eng.fp8_mode = False, None
                eng._cache_pool.clear()
    if rank == 0:
        cname = os.path.splitext(os.path.basename(str(args.config)))[0]
        adapters = "MLP adapters" if cname == "MAGMA_v1" else ("attention + MLP adapters" if cname == "MAGMA_v2" else "adapters per config")
        if args.train_only:
            # a short-lease multi-GPU record: BASELINE's first metric half alone ("train images/sec, whole node"); value / ms_per_step /
            # roofline are filled from the training leg below
            line = {"metric": f"train images/sec ({cname} training step, per-GPU batch {args.train_batch}, S = {model.seq_len}, whole job)",
                    "value": None, "unit": "images/s", "n_gpus": world, "steps": args.train_steps, "warmup": args.train_warmup,
                    "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "bf16", "data": "synthetic",
                    "config": {"workload": f"{cname} (CLIP RN50x16 + GPT-J-6B + {adapters}) bf16 training step: per-GPU batch "
                                           f"{args.train_batch} synthetic {args.res}x{args.res} image-caption pairs, S = {model.seq_len}, "
                                           f"adapters + image encoder + prefix trainable, clip + AdamW in the timed region",
                               "parallelism": f"dp{world}", "layers": model.lm.config.num_layers},
                    "train_only": True, "roofline": {"bound": "mfma"}}
        else:
            line = {"metric": f"generate tokens/sec ({cname}, batch-{B} images, {gen} new tokens, greedy)",
                    "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "bf16", "data": "synthetic",
                    "config": {"workload": f"{cname} (CLIP RN50x16 + GPT-J-6B + {adapters}) bf16 inference: batch {B} "
                                           f"{args.res}x{args.res} images + {args.prompt}-token prompt -> {gen} greedy tokens",
                               "parallelism": f"replicas x{world}", "layers": model.lm.config.num_layers,
                               "prefill_len": int(toks.shape[1] - gen)},
                    "per_rank_ms": gen_per_rank,          # each rank's own time per step (the headline divides by the MAX)
                    "roofline": roof}
            line["generate_sampled"] = gen_s
            line["generate_from_host"] = gen_h
            if gen8 is not None:
                line["generate_fp8"] = gen8
        line["train"] = None
    train = None
    if args.train_steps > 0:
        try:
            train = bench_train(model, args, rank, world, dev)
        except Exception as e:  # noqa: BLE001
            train = {"error": repr(e)[:300]}
    if rank == 0:
        line["train"] = train
        # second half of the metric in the SAME roofline object the driver parses: the training step is MFMA-bound, its
        # dominant kernel is the tile GEMM (measured live here, like the decode kernel above); flat copies for flat parsers
        if args.train_steps > 0 and line.get("roofline") is not None:
            try:
                mf = gemm_roofline(model, args, dev)
                line["roofline"]["train"] = mf
                line["roofline"].update({"mfma_kernel": mf["kernel"], "mfma_achieved_tflops": mf["achieved"], "mfma_peak_tflops": mf["peak"],
                                         "mfma_frac": mf["frac"]})
                full_ = (train or {}).get("full_S2048") or {}
                line["roofline"]["train_step_mfma_frac_executed"] = full_.get("mfma_frac_executed")
                if args.fp8 != "off":                 # BASELINE config[4]: the fp8 GEMM against the 5 PF/s dense fp8 peak
                    try:
                        line["roofline"]["train_fp8"] = gemm_roofline_fp8(model, args, dev)
                    except Exception as e:  # noqa: BLE001
                        line["roofline"]["train_fp8"] = {"error": repr(e)[:200]}
            except Exception as e:  # noqa: BLE001
                line["roofline"]["train"] = {"error": repr(e)[:200]}
        # data-parallel training throughput of the whole job (BASELINE metric, first half), next to the headline -- and, at the top
        # level where a flat parser keeps them, what a scaling curve is computed from: every rank's own step time, the gradient
        # exchange's exposed time and the exchange stream's busy time per step
        full = (train or {}).get("full_S2048") or {}
        line.update(flat_training_keys(train, world, args))
        if args.train_only:
            line["value"], line["ms_per_step"] = full.get("images_per_s"), full.get("ms_per_step")
            mf = (line.get("roofline") or {}).get("train")
            if isinstance(mf, dict) and "achieved" in mf:      # the dominant kernel of THIS workload in the contract's own keys
                line["roofline"].update({"bound": "mfma", "kernel": mf["kernel"], "achieved": mf["achieved"], "peak": mf["peak"],
                                         "unit": mf["unit"], "frac": mf["frac"], "traffic": None})
        elif world == 1 and args.layers is None:
            try:
                line["config1"] = config1_leg(model, dev)
            except Exception as e:  # noqa: BLE001
                line["config1"] = {"pass": False, "error": repr(e)[:300]}
        if not args.no_cpu_baseline:
            # rank 0 only, AFTER the last timed region; at N > 1 the other ranks are parked at the final barrier meanwhile (the
            # process group's timeout is 30 min), so the first multi-GPU record carries its CPU baseline too
            try:
                line["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"error": str(e)[:200]}
        if args.variants and world == 1 and args.layers is None:
            # BASELINE config[3] and the model-native resolution, driver-visible: each in its own try (the headline line must
            # survive), after the headline model has been released (the training step peaks at ~176 GB)
            cname = os.path.splitext(os.path.basename(str(args.config)))[0]
            try:
                line["generate_res384"] = variant_generate(model, args, dev, 384)
            except Exception as e:  # noqa: BLE001
                line["generate_res384"] = {"error": repr(e)[:300]}
            if cname == "MAGMA_v1":
                del model, eng
                try:
                    line["magma_v2"] = variant_v2(args, dev)
                except Exception as e:  # noqa: BLE001
                    line["magma_v2"] = {"error": repr(e)[:300]}
    if world == 1:
        _flush_c_stdio()
        print(json.dumps(line), flush=True)
        return
    # N ranks share one stdout.  The line has to be the LAST thing on it: RCCL (and gloo) write banners through C stdio, which is
    # block-buffered on a pipe and would otherwise drain at interpreter exit, behind the line.  So: leave the group together, every
    # rank pushes out what its C runtime still holds, ranks > 0 exit without running exit handlers, rank 0 prints last.
    torch.distributed.barrier()          # rank 0 measured the roofline objects alone: leave together
    torch.distributed.destroy_process_group()
    _flush_c_stdio()
    if rank != 0:
        os._exit(0)
    time.sleep(1.0)                      # the other ranks' last bytes reach the shared pipe first
    print(json.dumps(line), flush=True)
    sys.stderr.flush()
    os._exit(0)


def _flush_c_stdio():
    sys.stdout.flush()
    sys.stderr.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass


if __name__ == "__main__":
    main()
