"""the input vocabulary (rows of wte) and the output vocabulary (rows of the untied lm_head) are
independent.  The reference resizes the token embeddings to len(tokenizer) = 50258 (magma.py:50) on a model built with
50400 rows (language_model.py:19); whether the head follows is decided inside the un-vendored fork, so a published
checkpoint may carry either pair.  Product vs oracle at (1056, 1100) reduced -- loaded through load_checkpoint_state, which
must sniff both sizes -- and at (50258, 50400) full width on the head (GEMV, tile GEMM, loss)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def bf16_params(p):
    return {k: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k, v in p.items()}


def test_reduced_model_with_separate_vocabularies(dev):
    from magma_amd.testing import build_reduced_magma
    from oracle.model import OracleConfig, embed, generate_greedy, init_params, magma_forward
    cfg = OracleConfig.tiny(vocab_in=1056, vocab_out=1100)
    p = init_params(cfg, seed=23)
    model = build_reduced_magma(dev)                        # built at (1056, 1056): the checkpoint decides
    missing, unexpected = model.load_checkpoint_state(p)
    assert not unexpected
    assert model.lm.transformer.wte.weight.shape[0] == 1056 and model.lm.lm_head.weight.shape[0] == 1100
    assert model.lm.config.vocab_size == 1056 and model.lm.config.head_rows == 1100
    assert model.word_embedding is model.lm.transformer.wte
    model.eval()
    g = torch.Generator().manual_seed(5)
    images = torch.randn(2, 3, 64, 64, generator=g)
    ids = torch.randint(0, 1000, (2, 6), generator=g)
    emb_ref = embed(p, cfg, [images, ids])
    steps = 4
    ref_toks, ref_logits = generate_greedy(p, cfg, emb_ref, steps, stop_on_eos=False)
    _, bf_logits = generate_greedy(bf16_params(p), cfg, emb_ref.to(torch.bfloat16), steps, stop_on_eos=False)
    emb = emb_ref.to(torch.bfloat16).cuda()
    out = model.lm(inputs_embeds=emb, use_cache=True, cache_hint=steps)
    assert out.logits.shape[-1] == 1100
    assert rel(out.logits[:, -1], ref_logits[0]) <= 2 * rel(bf_logits[0], ref_logits[0]) + 2e-3
    past, S0 = out.past_key_values, emb.shape[1]
    for i in range(1, steps):
        out = model.lm(input_ids=ref_toks[:, S0 + i - 1: S0 + i].cuda(), use_cache=True, past_key_values=past)
        assert out.logits.shape[-1] == 1100
        assert rel(out.logits[:, -1], ref_logits[i]) <= 2 * max(rel(bf_logits[i], ref_logits[i]), 5e-3) + 2e-3
    # training-form forward: loss + full logits over the 1100-row head, labels drawn from the 1056-row input vocabulary
    S = model.seq_len
    caps = torch.randint(0, 1000, (2, S), generator=g)
    caps[:, 20:] = cfg.eos_token
    ref = magma_forward(p, cfg, images, caps)
    got = model(images.cuda(), caps.cuda(), return_logits=True)
    assert got.logits.shape == (2, S, 1100)
    assert abs(float(got.loss) - float(ref["loss"])) < 2e-2
    assert rel(got.logits, ref["logits"]) < 3e-2
    # the state dict round-trips with both sizes intact (save -> fresh model -> load)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    m2 = build_reduced_magma(dev)
    m2.load_checkpoint_state(sd)
    assert m2.lm.lm_head.weight.shape[0] == 1100 and m2.lm.transformer.wte.weight.shape[0] == 1056
    assert torch.equal(m2.lm.lm_head.weight, model.lm.lm_head.weight)


def test_resize_rules(dev):
    """resize_token_embeddings: HF's rule (head follows) by default, head kept on request, head on its own."""
    from magma_amd.language_model import GPTJConfig, get_gptj
    lm = get_gptj(device=dev, config=GPTJConfig(vocab_size=520, hidden_size=512, num_layers=1, num_heads=2, intermediate_size=1024,
                                               max_position_embeddings=64))
    w0, h0 = lm.transformer.wte.weight.clone(), lm.lm_head.weight.clone()
    lm.resize_token_embeddings(500)
    assert lm.transformer.wte.weight.shape[0] == 500 and lm.lm_head.weight.shape[0] == 500 and lm.config.vocab_out is None
    lm.resize_token_embeddings(480, resize_head=False)
    assert lm.transformer.wte.weight.shape[0] == 480 and lm.lm_head.weight.shape[0] == 500 and lm.config.head_rows == 500
    lm.resize_token_embeddings(new_head_rows=512)
    assert lm.transformer.wte.weight.shape[0] == 480 and lm.lm_head.weight.shape[0] == 512
    assert torch.equal(lm.transformer.wte.weight, w0[:480]) and torch.equal(lm.lm_head.weight[:500], h0[:500])


def test_full_width_head_at_50258_in_50400_out(dev):
    """(50258, 50400) at d = 4096: embedding gather from the 50258-row table, ln_f + lm_head over 50400 rows through the decode
    GEMV (M = 8), the tile GEMM (M = 64) and the loss head, against fp32 on the CPU."""
    from magma_amd import Magma
    from magma_amd.language_model import GPTJConfig
    torch.manual_seed(3)
    model = Magma("MAGMA_v1", device=dev, lm_config=GPTJConfig(num_layers=0, vocab_size=50258, vocab_out=50400))
    model.eval()
    lm = model.lm
    assert lm.transformer.wte.weight.shape == (50258, 4096) and lm.lm_head.weight.shape == (50400, 4096)
    with torch.no_grad():
        lm.transformer.ln_f.weight.normal_(1.0, 0.1)
        lm.transformer.ln_f.bias.normal_(0.0, 0.1)
    W, b = lm.lm_head.weight.float().cpu(), lm.lm_head.bias.float().cpu()
    g, be = lm.transformer.ln_f.weight.float().cpu(), lm.transformer.ln_f.bias.float().cpu()
    wte = lm.transformer.wte.weight.float().cpu()

    def ref_logits(x):
        return torch.nn.functional.layer_norm(x, (4096,), g, be, 1e-5) @ W.t() + b

    gen = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 50258, (8, 9), generator=gen)
    ids[0, -1] = 50257                                              # last row of the input table
    out = lm(input_ids=ids.cuda(), use_cache=True, cache_hint=4)     # zero blocks: logits = head(ln_f(wte[ids]))
    assert out.logits.shape == (8, 1, 50400)
    ref = ref_logits(wte[ids[:, -1]])
    assert rel(out.logits[:, 0], ref) < 6e-3
    assert bool((out.logits[:, 0].argmax(-1).cpu() == ref.argmax(-1)).all()) or rel(out.logits[:, 0], ref) < 3e-3
    # the LayerNorm-folded weight-streaming GEMV of the token step (M = 8) on the 50400-row head
    from magma_amd import ops
    eng = lm.engine
    eng._ensure_decode_packs()
    nxt = torch.randint(0, 50258, (8,), generator=gen)
    x = lm.transformer.wte.weight[nxt.cuda()].contiguous()
    lg = torch.empty(8, eng.Vp, dtype=torch.float32, device=dev)
    ops.gemm_skinny(x, eng.head_dec, out=lg, ln_fold=(eng.head_dec.colsum, eng.d, eng.eps))
    assert eng.V == 50400 and rel(lg[:, :50400], ref_logits(wte[nxt])) < 6e-3
    full = lm(input_ids=ids.cuda())                                  # (8, 9, 50400) through the tile GEMM
    assert full.logits.shape == (8, 9, 50400)
    assert rel(full.logits, ref_logits(wte[ids])) < 8e-3
    # loss with labels in the OUTPUT vocabulary's upper range (ids >= 50258 exist only there)
    labels = torch.randint(50258, 50400, (8, 9), generator=gen)
    o = lm(input_ids=ids.cuda(), labels=labels.cuda())
    lg = ref_logits(wte[ids])[:, :-1].reshape(-1, 50400)
    ref_loss = torch.nn.functional.cross_entropy(lg, labels[:, 1:].reshape(-1))
    assert abs(float(o.loss) - float(ref_loss)) < 2e-2
