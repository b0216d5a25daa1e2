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


def test_vit_oracle_matches_hf_clip_vision():
    """oracle.vit_encoder_fwd (CLIP VisionTransformer restated from the published architecture, openai/CLIP parameter
    names) against HF CLIPVisionModelWithProjection -- an independent implementation of the same model -- on a reduced
    configuration: patch conv, class / positional embeddings, pre-LN blocks with QuickGELU, ln_post, projection."""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from oracle.model import ViTConfig, init_vit_params, vit_encoder_fwd
    v = ViTConfig(width=64, layers=3, heads=4, patch=8, resolution=32, out_dim=24)
    p = init_vit_params(v, seed=5, prefix="")
    hf = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=64, intermediate_size=256, num_hidden_layers=3,
                                                        num_attention_heads=4, image_size=32, patch_size=8, projection_dim=24,
                                                        hidden_act="quick_gelu", layer_norm_eps=1e-5, attn_implementation="eager"))
    hf.eval()
    sd = hf.state_dict()
    m = {"vision_model.embeddings.patch_embedding.weight": p["conv1.weight"],
         "vision_model.embeddings.class_embedding": p["class_embedding"],
         "vision_model.embeddings.position_embedding.weight": p["positional_embedding"],
         "vision_model.pre_layrnorm.weight": p["ln_pre.weight"], "vision_model.pre_layrnorm.bias": p["ln_pre.bias"],
         "vision_model.post_layernorm.weight": p["ln_post.weight"], "vision_model.post_layernorm.bias": p["ln_post.bias"],
         "visual_projection.weight": p["proj"].t().contiguous()}
    for i in range(v.layers):
        b, h = f"transformer.resblocks.{i}.", f"vision_model.encoder.layers.{i}."
        wq, wk, wv = p[b + "attn.in_proj_weight"].chunk(3, 0)
        bq, bk, bv = p[b + "attn.in_proj_bias"].chunk(3, 0)
        m.update({h + "self_attn.q_proj.weight": wq, h + "self_attn.k_proj.weight": wk, h + "self_attn.v_proj.weight": wv,
                  h + "self_attn.q_proj.bias": bq, h + "self_attn.k_proj.bias": bk, h + "self_attn.v_proj.bias": bv,
                  h + "self_attn.out_proj.weight": p[b + "attn.out_proj.weight"], h + "self_attn.out_proj.bias": p[b + "attn.out_proj.bias"],
                  h + "layer_norm1.weight": p[b + "ln_1.weight"], h + "layer_norm1.bias": p[b + "ln_1.bias"],
                  h + "layer_norm2.weight": p[b + "ln_2.weight"], h + "layer_norm2.bias": p[b + "ln_2.bias"],
                  h + "mlp.fc1.weight": p[b + "mlp.c_fc.weight"], h + "mlp.fc1.bias": p[b + "mlp.c_fc.bias"],
                  h + "mlp.fc2.weight": p[b + "mlp.c_proj.weight"], h + "mlp.fc2.bias": p[b + "mlp.c_proj.bias"]})
    missing = [k for k in sd if k not in m and "position_ids" not in k]
    assert not missing, missing
    hf.load_state_dict({k: m.get(k, sd[k]) for k in sd})
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = hf(pixel_values=x).image_embeds
        got = vit_encoder_fwd(p, v, x, prefix="")
    assert got.shape == ref.shape == (2, 24)
    assert torch.allclose(got, ref, atol=2e-5, rtol=1e-4), float((got - ref).abs().max())
