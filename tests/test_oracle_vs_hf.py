"""Pins the oracle's GPT-J arithmetic to the one independent statement of the
published algorithm available offline: HF ``GPTJForCausalLM`` (the reference's
own implementation lives in an un-vendored fork).  CPU, fp32."""
import pytest
import torch

from oracle import model as O

transformers = pytest.importorskip("transformers")


def _hf_model(cfg):
    from transformers import GPTJConfig, GPTJForCausalLM
    hc = GPTJConfig(vocab_size=cfg.vocab_out, n_positions=cfg.n_positions, n_embd=cfg.d_model, n_layer=cfg.n_layer,
                    n_head=cfg.n_head, rotary_dim=cfg.rotary_dim, n_inner=cfg.d_ff, activation_function="gelu_new",
                    resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0, layer_norm_epsilon=cfg.ln_eps,
                    tie_word_embeddings=False)
    hc._attn_implementation = "eager"
    return GPTJForCausalLM(hc).eval()


def _copy_params(hf, p, cfg):
    sd = {"transformer.wte.weight": p["lm.transformer.wte.weight"],
          "transformer.ln_f.weight": p["lm.transformer.ln_f.weight"], "transformer.ln_f.bias": p["lm.transformer.ln_f.bias"],
          "lm_head.weight": p["lm.lm_head.weight"], "lm_head.bias": p["lm.lm_head.bias"]}
    for i in range(cfg.n_layer):
        h, t = f"lm.transformer.h.{i}.", f"transformer.h.{i}."
        sd[t + "ln_1.weight"], sd[t + "ln_1.bias"] = p[h + "ln_1.weight"], p[h + "ln_1.bias"]
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[t + f"attn.{n}.weight"] = p[O.attn_prefix(cfg, i) + n + ".weight"]
        mp = O.mlp_prefix(cfg, i)
        sd[t + "mlp.fc_in.weight"], sd[t + "mlp.fc_in.bias"] = p[mp + "c_fc.weight"], p[mp + "c_fc.bias"]
        sd[t + "mlp.fc_out.weight"], sd[t + "mlp.fc_out.bias"] = p[mp + "c_proj.weight"], p[mp + "c_proj.bias"]
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not [m for m in missing if "bias" not in m.split(".")[-1] or "attn" not in m], missing
    assert not unexpected, unexpected


@pytest.fixture(scope="module")
def setup():
    cfg = O.OracleConfig.tiny(n_layer=2, n_head=1, d_ff=512, vocab_in=96, vocab_out=96, n_positions=64,
                              mlp_adapter_hidden=0, attn_adapter_hidden=0)
    p = O.init_params(cfg, seed=5, lm_std=0.08)
    hf = _hf_model(cfg)
    _copy_params(hf, p, cfg)
    return cfg, p, hf


def test_logits_and_loss_match_hf(setup):
    cfg, p, hf = setup
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, cfg.vocab_in, (2, 19), generator=g)
    labels = ids.clone()
    labels[:, :4] = -100
    labels[1, 12:] = -100
    with torch.no_grad():
        ref = hf(input_ids=ids, labels=labels)
        got = O.lm_forward(p, cfg, input_ids=ids, labels=labels)
    assert torch.allclose(got["logits"], ref.logits, atol=2e-4, rtol=2e-4), float((got["logits"] - ref.logits).abs().max())
    assert abs(float(got["loss"]) - float(ref.loss)) < 1e-4


def test_cached_decode_matches_hf(setup):
    cfg, p, hf = setup
    g = torch.Generator().manual_seed(1)
    emb = torch.randn(2, 7, cfg.d_model, generator=g) * 0.5
    with torch.no_grad():
        ref_toks = []
        out = hf(inputs_embeds=emb, use_cache=True)
        past = out.past_key_values
        nxt = out.logits[:, -1].argmax(-1, keepdim=True)
        ref_logits = [out.logits[:, -1]]
        for _ in range(4):
            ref_toks.append(nxt)
            out = hf(input_ids=nxt, past_key_values=past, use_cache=True)
            past = out.past_key_values
            ref_logits.append(out.logits[:, -1])
            nxt = out.logits[:, -1].argmax(-1, keepdim=True)
        ref_toks.append(nxt)
    toks, step_logits = O.generate_greedy(p, cfg, emb, 5, stop_on_eos=False)
    for a, b in zip(step_logits, ref_logits):
        assert torch.allclose(a, b, atol=3e-4, rtol=3e-4), float((a - b).abs().max())
    assert torch.equal(toks[:, 7:], torch.cat(ref_toks, dim=1))


def test_adapter_placement_sequential_after_mlp():
    """MLP adapter = Sequential(mlp, Adapter) (reference magma.py:143-149): the
    oracle with adapters equals the no-adapter block whose MLP output m is
    replaced by m + adapter(m)."""
    cfg = O.OracleConfig.tiny(n_layer=1, n_head=1, d_ff=512, vocab_in=96, vocab_out=96, mlp_adapter_hidden=64)
    p = O.init_params(cfg, seed=2)
    x = torch.randn(1, 5, cfg.d_model)
    ln = torch.nn.functional.layer_norm(x, (cfg.d_model,), p["lm.transformer.h.0.ln_1.weight"], p["lm.transformer.h.0.ln_1.bias"], cfg.ln_eps)
    mp = O.mlp_prefix(cfg, 0)
    m = torch.nn.functional.linear(O.gelu_new(torch.nn.functional.linear(ln, p[mp + "c_fc.weight"], p[mp + "c_fc.bias"])), p[mp + "c_proj.weight"], p[mp + "c_proj.bias"])
    want = O.adapter_fwd(p, "lm.transformer.h.0.mlp.1.adapter.", m)
    assert torch.allclose(O.mlp_fwd(p, cfg, 0, ln), want, atol=1e-6)
