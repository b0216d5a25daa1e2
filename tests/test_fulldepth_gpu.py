"""Parity at full DEPTH and in the configurations the one-block MAGMA_v1 tests (tests/test_fullwidth_gpu.py) do not reach:

  (a) MAGMA_v1 at 28 blocks, d 4096, ff 16384, V 50258 -- the model the headline measures: prefill (B = 2, S0 = 57) and 8
      cached steps against the fp32 CPU oracle, logits under the 2 x eager-bf16 criterion, and FREE-RUNNING greedy ids
      through generate() exactly equal to the oracle's on margin-controlled inputs (reference magma/sampling.py:81-97);
  (b) MAGMA_v2 (attention AND mlp adapters, downsample 8; reference configs/MAGMA_v2.yml:4, magma/adapters.py:95-116) at
      full width: prefill, the five-launch decode block (concatenated up-projection GEMV), gradients at S = 2048;
  (c) the model-native 384^2 image -> 144 prefix tokens -> prefill at S0 = 152 (reference magma/image_prefix.py:13,
      magma/magma.py:238-276) at full width: M = 8 x 152 rows take other tile / split-K choices than M = 456;
  (d) W8A16 decode and the fp8 ('all') prefill at full width against the fp32 oracle evaluated on the DEQUANTISED weights
      (BASELINE config[4]).

Tolerance: err(HIP bf16, oracle fp32) <= 2 x err(oracle in bf16 on PyTorch CPU, oracle fp32) + floor (rel-L2,
floors stated per assert); greedy token ids EXACT (H2: wherever the oracle's own top-1 margin exceeds bf16 noise)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fullwidth_common as F  # noqa: E402

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
TINY_TRUNK = dict(enc_width=16, enc_layers=(1, 1, 2, 1))      # the depth / variant tests are about the LM side


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def bf16_params(p):
    return {k: (v.to(BF16) if v.is_floating_point() else v) for k, v in p.items()}


def check(err_hip, err_bf16, what, floor=2e-3):
    print(f"{what}: HIP err {err_hip:.3e}, eager-bf16 err {err_bf16:.3e}")
    assert err_hip <= 2.0 * err_bf16 + floor, f"{what}: HIP err {err_hip:.3e} vs eager-bf16 err {err_bf16:.3e}"


def build(dev, cfg, params, resolution=64, **kw):
    from magma_amd.testing import build_reduced_magma
    model = build_reduced_magma(dev, n_layer=cfg.n_layer, n_head=16, d_ff=16384, vocab=50258, n_positions=cfg.n_positions,
                                enc_width=cfg.enc_width, enc_layers=cfg.enc_layers, resolution=resolution, **kw)
    missing, unexpected = model.load_checkpoint_state(params)
    assert not unexpected and not missing, (missing[:4], unexpected[:4])
    model.eval()
    return model


def margin_safe(ref_logits):
    top2 = torch.topk(ref_logits, 2, dim=-1).values
    return (top2[:, 0] - top2[:, 1]) > F.TEST_MARGIN * ref_logits.std(dim=-1)


# ------------------------------------------------------------------------------------------------ (a) 28 blocks
@pytest.fixture(scope="module")
def v1_28(dev):
    """MAGMA_v1 at all 28 blocks (5.9 B values drawn once for both 28-block tests)."""
    cfg = F.full_width_config(n_layer=28, **TINY_TRUNK)
    params = F.full_depth_params(cfg)
    model = build(dev, cfg, params)
    return cfg, params, model


def test_magma_v1_at_28_blocks(v1_28):
    """The headline model, every block of it: error accumulation through 28 parallel-residual blocks, the 28-layer KV cache,
    the captured 115-launch token step."""
    from oracle.model import lm_forward
    cfg, params, model = v1_28
    lm = F.lm_only(params)
    emb = F.greedy_inputs(cfg, F.FULLDEPTH_INPUT_SEED)
    steps, S0 = F.FULLDEPTH_STEPS, F.PREFILL_LEN
    with torch.no_grad():
        ref_toks, margins, ref_logits = F.oracle_greedy_margins(lm, cfg, emb, steps)
        assert min(margins) > F.TEST_MARGIN, f"fixture lost its margin on this host: {margins}"
        # the error yardstick: the same graph, same tokens, in bf16 on PyTorch CPU
        lmb = bf16_params(lm)
        rb = lm_forward(lmb, cfg, inputs_embeds=emb.to(BF16))
        e_bf = [rel(rb["logits"][:, -1], ref_logits[0])]
        pastb = rb["past_key_values"]
        for i in range(1, steps):
            rb = lm_forward(lmb, cfg, input_ids=ref_toks[:, S0 + i - 1: S0 + i], past=pastb)
            pastb = rb["past_key_values"]
            e_bf.append(rel(rb["logits"][:, -1], ref_logits[i]))
        del lmb, rb, pastb
        # HIP: teacher-forced logits step by step ...
        out = model.lm(inputs_embeds=emb.to(BF16).cuda(), use_cache=True, cache_hint=steps + 8)
        check(rel(out.logits[:, -1], ref_logits[0]), e_bf[0], "28 blocks: prefill logits (last row)", floor=3e-3)
        cache = out.past_key_values
        for i in range(1, steps):
            o = model.lm(input_ids=ref_toks[:, S0 + i - 1: S0 + i].cuda(), use_cache=True, past_key_values=cache)
            check(rel(o.logits[:, -1], ref_logits[i]), e_bf[i], f"28 blocks: cached step {i} logits", floor=3e-3)
            assert torch.equal(o.next_token.cpu(), ref_toks[:, S0 + i])
        # ... and free-running through the public generate() (decode graph, in-graph argmax, tokens fed back on the device)
        toks = model.generate(emb.to(BF16).cuda(), max_steps=steps, temperature=0.0, decode=False, stop_on_eos=False).cpu()
    assert toks.shape == ref_toks.shape == (F.GREEDY_B, S0 + steps)
    assert torch.equal(toks, ref_toks), (toks[:, S0:], ref_toks[:, S0:], margins)


def test_training_engine_loss_at_28_blocks_s2048(v1_28, dev):
    """The TRAINING engine's forward (taped activations, the 256x256 GEMMs at M = 2048 rows, flash attention at S = 2048 with the
    LSE written for the backward, rotary split that also emits q^T / k^T, the loss head on the target rows) through all 28 blocks
    at the sequence length of BASELINE config[2], against the fp32 oracle forward (reference magma.py:238-276): the loss and the
    fp32 logits of every row that carries a target, under the 2 x eager-bf16 criterion.  The engine runs all 2048 positions; the CPU oracle evaluates the
    same loss / target-row logits on the window that carries them (fullwidth_common.oracle_window: causal mask + masked loss)."""
    from magma_amd.train_engine import MagmaEngine
    from oracle.model import magma_forward
    cfg, params, model = v1_28
    model.config.gradient_accumulation_steps = 1
    eng = MagmaEngine(model)
    eng.train()
    B, S = 1, 2048
    P = (64 // 32) ** 2
    g = torch.Generator().manual_seed(29)
    images = torch.randn(B, 3, 64, 64, generator=g).to(BF16).float()
    caps = torch.full((B, S), cfg.eos_token, dtype=torch.int64)
    n_tok = 96
    caps[0, :n_tok] = torch.randint(0, 50256, (n_tok,), generator=g)
    mask = (torch.rand(B, P, cfg.d_model, generator=g) < 0.9).float() / 0.9
    with torch.no_grad():
        win = F.oracle_window(caps, P, cfg.eos_token)       # the oracle's view: same loss / target-row logits (fullwidth_common.oracle_window)
        ref = magma_forward(params, cfg, images, win, dropout_mask=mask)
        pb = bf16_params(params)
        rb = magma_forward(pb, cfg, images.to(BF16), win, dropout_mask=mask.to(BF16))
        del pb
    labels = ref["labels"]
    rows = (labels[0, 1:] != -100).nonzero().squeeze(1)           # positions whose NEXT token carries a label
    assert rows.numel() == n_tok + 1                               # the caption tokens + the first eos (reference utils.py:334-364)
    ref_lg, bf_lg = ref["logits"][0, rows].float(), rb["logits"][0, rows].float()
    loss_ref, loss_bf = float(ref["loss"]), float(rb["loss"])
    del ref, rb
    out = eng(images.to(dev), caps.to(dev), dropout_mask=mask.to(dev))
    assert torch.equal(out.target_rows.cpu(), rows)
    e_hip, e_bf = rel(out.target_logits, ref_lg), rel(bf_lg, ref_lg)
    print(f"28-block training forward: loss HIP {float(out.loss):.5f} oracle {loss_ref:.5f} eager-bf16 {loss_bf:.5f}; "
          f"target-row logits HIP {e_hip:.3e} eager-bf16 {e_bf:.3e}")
    assert abs(float(out.loss) - loss_ref) <= 2 * abs(loss_bf - loss_ref) + 3e-3 * abs(loss_ref), (float(out.loss), loss_ref, loss_bf)
    check(e_hip, e_bf, "28 blocks, S = 2048: target-row logits of the training forward", floor=3e-3)
    # the backward runs on this tape (28 blocks of saved activations) and leaves finite, non-zero adapter gradients
    eng.backward(out.loss)
    ad = model.lm.transformer.h[0].mlp[1].adapter[0].weight
    gn = float(eng.grad_of(ad).float().norm())
    assert gn > 0 and gn == gn
    model.zero_grad(set_to_none=True)
    for grp in eng.groups:
        grp.grad.zero_()


# ------------------------------------------------------------------------------------------------ (b) MAGMA_v2
@pytest.fixture(scope="module")
def v2(dev):
    cfg = F.full_width_config(mlp_adapter_hidden=512, attn_adapter_hidden=512, **TINY_TRUNK)
    params = F.full_width_params(cfg)
    model = build(dev, cfg, params, mlp_factor=8, attn_factor=8)
    return cfg, params, model


def test_magma_v2_prefill_and_decode_block(v2):
    """B = 8, S0 = 57, then cached steps through the five-launch MAGMA_v2 decode block (attention adapter after out_proj,
    both up-projections as ONE GEMV over the concatenated bottlenecks -- engine._adapter_up_cat)."""
    from oracle.model import lm_forward
    cfg, p, model = v2
    lm, lmb = F.lm_only(p), bf16_params(F.lm_only(p))
    emb = F.greedy_inputs(cfg, seed=2468, B=8)
    steps = 4
    with torch.no_grad():
        r = lm_forward(lm, cfg, inputs_embeds=emb)
        rb = lm_forward(lmb, cfg, inputs_embeds=emb.to(BF16))
        out = model.lm(inputs_embeds=emb.to(BF16).cuda(), use_cache=True, cache_hint=steps + 8)
        check(rel(out.logits[:, -1], r["logits"][:, -1]), rel(rb["logits"][:, -1], r["logits"][:, -1]), "v2 prefill logits")
        past, pastb, cache = r["past_key_values"], rb["past_key_values"], out.past_key_values
        tok = r["logits"][:, -1].argmax(-1, keepdim=True)
        n_safe = 0
        for i in range(steps):      # eager step, graph capture, replays
            r = lm_forward(lm, cfg, input_ids=tok, past=past)
            rb = lm_forward(lmb, cfg, input_ids=tok, past=pastb)
            o = model.lm(input_ids=tok.cuda(), use_cache=True, past_key_values=cache)
            ref = r["logits"][:, -1]
            check(rel(o.logits[:, -1], ref), rel(rb["logits"][:, -1], ref), f"v2 cached step {i} logits")
            safe = margin_safe(ref)
            n_safe += int(safe.sum())
            assert bool((o.next_token.cpu()[safe] == ref.argmax(-1)[safe]).all())
            past, pastb = r["past_key_values"], rb["past_key_values"]
            tok = ref.argmax(-1, keepdim=True)
        assert n_safe >= 0.7 * 8 * steps
    eng = model.lm.engine
    assert eng._adapter_up_cat(eng.layers[0]) is not None, "the v2 block did not take the fused up-projection path"


def test_magma_v2_gradients_s2048(v2, dev):
    """config[3] training shapes at full width: loss and the gradient of every trainable tensor (attention adapter, mlp
    adapter, trunk, prefix) against torch.autograd through the fp32 oracle; per tensor err <= 2 x bf16-autograd + 1e-2."""
    from magma_amd.train_engine import MagmaEngine
    from oracle.model import magma_forward
    cfg, params, model = v2
    model.config.gradient_accumulation_steps = 1
    eng = MagmaEngine(model)
    eng.train()
    B, S = 2, 2048
    P = (64 // 32) ** 2            # 64^2 images through the reduced trunk: a 2 x 2 token grid
    g = torch.Generator().manual_seed(17)
    images = torch.randn(B, 3, 64, 64, generator=g).to(BF16).float()
    caps = torch.full((B, S), cfg.eos_token, dtype=torch.int64)
    caps[0, :45] = torch.randint(0, 50256, (45,), generator=g)
    caps[1, :23] = torch.randint(0, 50256, (23,), generator=g)
    mask = (torch.rand(B, P, cfg.d_model, generator=g) < 0.9).float() / 0.9
    names = [k for k in params if (".adapter." in k or k.startswith("image_prefix.")) and "running_" not in k]

    def oracle(dtype):
        p = {k: (v.detach().to(dtype).clone() if v.is_floating_point() else v) for k, v in params.items()}
        for k in names:
            p[k].requires_grad_(True)
        out = magma_forward(p, cfg, images.to(dtype), F.oracle_window(caps, P, cfg.eos_token), dropout_mask=mask.to(dtype))
        out["loss"].backward()
        return float(out["loss"].detach()), {k: p[k].grad.float() for k in names}

    loss_ref, g_ref = oracle(torch.float32)
    loss_bf, g_bf = oracle(BF16)
    out = eng(images.to(dev), caps.to(dev), dropout_mask=mask.to(dev))
    assert abs(float(out.loss) - loss_ref) <= 2 * abs(loss_bf - loss_ref) + 3e-3 * abs(loss_ref), (float(out.loss), loss_ref, loss_bf)
    eng.backward(out.loss)
    name_of = {id(p): n for n, p in model.named_parameters()}
    seen, bad, worst = set(), [], []
    for grp in eng.groups:
        for p in grp.params:
            n = name_of[id(p)]
            n = "lm." + n if n.startswith("transformer.") else n
            if n in seen or n not in g_ref:
                continue
            seen.add(n)
            got, ref = eng.grad_of(p).float().cpu().reshape(-1), g_ref[n].reshape(-1)
            e_hip, e_bf = rel(got, ref), rel(g_bf[n].reshape(-1), ref)
            worst.append((e_hip - 2 * e_bf, n, e_hip, e_bf))
            if e_hip > 2 * e_bf + 1e-2:
                bad.append((n, e_hip, e_bf))
    worst.sort(reverse=True)
    print("v2 loss", float(out.loss), loss_ref, loss_bf, "| worst:", [(n, f"{a:.2e}", f"{b:.2e}") for _, n, a, b in worst[:6]])
    assert len(seen) == len(g_ref), (len(seen), len(g_ref))
    assert any(".attn.adapter." in n for n in seen) and any(".mlp.1.adapter." in n for n in seen)
    assert not bad, bad[:8]
    model.zero_grad(set_to_none=True)


# ------------------------------------------------------------------------------------------------ (c) 384^2, S0 = 152
def test_native_resolution_384_prefix_and_prefill(dev):
    """One 384^2 image through the full RN50x16 trunk (144 positions x 3072 channels, reference image_prefix.py:13,20) and
    the prefix; then BASELINE config[1] at the model-native prompt length: B = 8, S0 = 144 + 8 = 152 (M = 1216 rows)."""
    from oracle.model import image_prefix_fwd, lm_forward
    cfg = F.full_width_config()
    p = F.full_width_params(cfg)
    model = build(dev, cfg, p, resolution=384)
    g = torch.Generator().manual_seed(11)
    img = torch.randn(1, 3, 384, 384, generator=g).to(BF16).float()
    lm, lmb = F.lm_only(p), bf16_params(F.lm_only(p))
    with torch.no_grad():
        ref = image_prefix_fwd(p, cfg, img)
        eb = rel(image_prefix_fwd(bf16_params(p), cfg, img.to(BF16)), ref)
        got = model.image_prefix(img.cuda())
        assert got.shape == ref.shape == (1, 144, 4096)
        check(rel(got, ref), eb, "ImagePrefix @384", floor=5e-3)
        emb = F.greedy_inputs(cfg, seed=1357, B=8, S0=152)
        r = lm_forward(lm, cfg, inputs_embeds=emb)
        rb = lm_forward(lmb, cfg, inputs_embeds=emb.to(BF16))
        out = model.lm(inputs_embeds=emb.to(BF16).cuda(), use_cache=True, cache_hint=16)
        check(rel(out.logits[:, -1], r["logits"][:, -1]), rel(rb["logits"][:, -1], r["logits"][:, -1]), "S0=152 prefill logits")
        full = model.lm(inputs_embeds=emb.to(BF16).cuda()).logits
        check(rel(full, r["logits"]), rel(rb["logits"], r["logits"]), "S0=152 full (B,S,V) logits", floor=4e-3)
        past, pastb, cache = r["past_key_values"], rb["past_key_values"], out.past_key_values
        tok = r["logits"][:, -1].argmax(-1, keepdim=True)
        for i in range(3):
            r = lm_forward(lm, cfg, input_ids=tok, past=past)
            rb = lm_forward(lmb, cfg, input_ids=tok, past=pastb)
            o = model.lm(input_ids=tok.cuda(), use_cache=True, past_key_values=cache)
            ref_l = r["logits"][:, -1]
            check(rel(o.logits[:, -1], ref_l), rel(rb["logits"][:, -1], ref_l), f"ctx 152+{i} cached step logits")
            safe = margin_safe(ref_l)
            assert bool((o.next_token.cpu()[safe] == ref_l.argmax(-1)[safe]).all())
            past, pastb = r["past_key_values"], rb["past_key_values"]
            tok = ref_l.argmax(-1, keepdim=True)


# ------------------------------------------------------------------------------------------------ (d) e4m3 weights
def _no_ln_bias(params):
    """LayerNorm biases zeroed: with beta = 0 the LayerNorm fold of the decode operands (W' = W * gamma quantised per output
    channel, b' = b + W beta) is EXACTLY 'the oracle on the weights dequant(W') / gamma' -- no bias correction term."""
    p = dict(params)
    for k in p:
        if k.endswith("ln_1.bias") or k.endswith("ln_f.bias"):
            p[k] = torch.zeros_like(p[k])
    return p


def test_w8a16_decode_vs_oracle_on_dequantised_weights(dev):
    """W8A16 token step at full width (MAGMA_DECODE_W8: e4m3 weights, per-output-channel scales, widened to bf16 in
    registers) against the fp32 oracle evaluated on the dequantised weights, from the SAME bf16 prefill cache.  What is
    left between the two is bf16 activation rounding only: 2 x eager-bf16 criterion."""
    from oracle.model import attn_prefix, lm_forward, mlp_adapter_prefix, mlp_prefix
    cfg = F.full_width_config(**TINY_TRUNK)
    p = _no_ln_bias(F.full_width_params(cfg))
    model = build(dev, cfg, p)
    eng = model.lm.engine
    lm = F.lm_only(p)
    emb = F.greedy_inputs(cfg, seed=97, B=8)
    try:
        eng.decode_w8 = True
        with torch.no_grad():
            r0 = lm_forward(lm, cfg, inputs_embeds=emb)
            out = model.lm(inputs_embeds=emb.to(BF16).cuda(), use_cache=True, cache_hint=8)     # prefill: bf16 weights
            tok = r0["logits"][:, -1].argmax(-1, keepdim=True)
            o = model.lm(input_ids=tok.cuda(), use_cache=True, past_key_values=out.past_key_values)
            ly = eng.layers[0]
            assert getattr(ly, "w8", None) is not None, "the W8A16 operands were not built: the e4m3 path did not run"
            # the oracle's weights := what the kernels multiply by
            d, d3 = cfg.d_model, 3 * cfg.d_model
            q = dict(lm)
            gam = lm["lm.transformer.h.0.ln_1.weight"]
            w_in = ly.w8.dec_in.dequant().cpu() / gam[None, :]
            ap, mp, adp = attn_prefix(cfg, 0), mlp_prefix(cfg, 0), mlp_adapter_prefix(cfg, 0)
            q[ap + "q_proj.weight"], q[ap + "k_proj.weight"], q[ap + "v_proj.weight"] = w_in[:d], w_in[d:2 * d], w_in[2 * d:d3]
            q[mp + "c_fc.weight"] = w_in[d3:]
            q[ap + "out_proj.weight"] = ly.w8.out.dequant().cpu()
            q[mp + "c_proj.weight"] = ly.w8.fc_out.dequant().cpu()
            q[adp + "0.weight"] = ly.w8.mlp_adapter[0].dequant().cpu()
            q[adp + "2.weight"] = ly.w8.mlp_adapter[1].dequant().cpu()
            q["lm.lm_head.weight"] = eng.head_w8.dequant().cpu()[: cfg.vocab_out] / lm["lm.transformer.ln_f.weight"][None, :]
            r = lm_forward(q, cfg, input_ids=tok, past=r0["past_key_values"])
            rb = lm_forward(bf16_params(q), cfg, input_ids=tok, past=[(k.to(BF16), v.to(BF16)) for k, v in r0["past_key_values"]])
            ref = r["logits"][:, -1]
            check(rel(o.logits[:, -1], ref), rel(rb["logits"][:, -1], ref), "W8A16 step logits vs dequantised oracle", floor=3e-3)
            unq = lm_forward(lm, cfg, input_ids=tok, past=r0["past_key_values"])["logits"][:, -1]
            print("e4m3 weight quantisation itself moves the logits by", rel(ref, unq))
            assert rel(ref, unq) > 5 * rel(o.logits[:, -1], ref), "the comparison would not notice unquantised weights"
            safe = margin_safe(ref)
            assert bool((o.next_token.cpu()[safe] == ref.argmax(-1)[safe]).all())
    finally:
        eng.decode_w8 = False
        eng._cache_pool.clear()


@pytest.mark.parametrize("scaling", ["row", "mx"])
def test_fp8_prefill_vs_oracle_on_dequantised_weights(dev, scaling):
    """(scaling "mx": OCP MX block scales -- one E8M0 per 32 K-elements of activations and weights, applied by the MFMA.)
    MAGMA_FP8=all prefill at full width (qkv, out_proj, fc_in, fc_out and the adapter projections on the MX-rate fp8
    MFMA) against the fp32 oracle on the dequantised e4m3 weights.  The oracle keeps fp32 ACTIVATIONS, the kernels quantise
    them per row to e4m3 (3 mantissa bits, the same format as the weights) in front of every projection.  Stated tolerance,
    self-calibrating: the activation quantisation may move the logits by at most 1.5 x what the e4m3 WEIGHT quantisation
    itself moves them (rel-L2 between the dequantised and the unquantised oracle; ~4e-2 each on this model), and the fp8
    path must sit closer to its own dequantised oracle than to the unquantised one."""
    from oracle.model import attn_prefix, lm_forward, mlp_adapter_prefix, mlp_prefix
    cfg = F.full_width_config(**TINY_TRUNK)
    p = F.full_width_params(cfg)
    model = build(dev, cfg, p)
    eng = model.lm.engine
    lm = F.lm_only(p)
    emb = F.greedy_inputs(cfg, seed=31, B=8)
    try:
        eng.fp8_mode, eng.fp8_scaling = "all", scaling
        with torch.no_grad():
            got = model.lm(inputs_embeds=emb.to(BF16).cuda()).logits.float().cpu()
            packs = eng.layers[0].fp8
            from magma_amd import ops
            assert all(isinstance(v, ops.PackedLinearMX) == (scaling == "mx") for v in packs.values())
            assert set(packs) >= {"qkv", "out", "fc_in", "fc_out", "mlp_dn", "mlp_up"}, sorted(packs)
            d = cfg.d_model
            q = dict(lm)
            ap, mp, adp = attn_prefix(cfg, 0), mlp_prefix(cfg, 0), mlp_adapter_prefix(cfg, 0)
            w = packs["qkv"].dequant().cpu()
            q[ap + "q_proj.weight"], q[ap + "k_proj.weight"], q[ap + "v_proj.weight"] = w[:d], w[d:2 * d], w[2 * d:3 * d]
            q[ap + "out_proj.weight"] = packs["out"].dequant().cpu()
            q[mp + "c_fc.weight"] = packs["fc_in"].dequant().cpu()
            q[mp + "c_proj.weight"] = packs["fc_out"].dequant().cpu()
            q[adp + "0.weight"] = packs["mlp_dn"].dequant().cpu()
            q[adp + "2.weight"] = packs["mlp_up"].dequant().cpu()
            ref = lm_forward(q, cfg, inputs_embeds=emb)["logits"]
            unq = lm_forward(lm, cfg, inputs_embeds=emb)["logits"]
        e_deq, e_unq = rel(got, ref), rel(got, unq)
        print(f"fp8 'all' ({scaling} scales) prefill logits: vs dequantised oracle {e_deq:.3e}, vs unquantised oracle {e_unq:.3e}")
        assert torch.isfinite(got).all()
        e_w = rel(ref, unq)
        print(f"e4m3 weight quantisation alone moves the logits by {e_w:.3e}")
        assert e_deq <= 1.5 * e_w, (e_deq, e_w)
        assert e_deq < e_unq
    finally:
        eng.fp8_mode, eng.fp8_scaling = None, "row"
