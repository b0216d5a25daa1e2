"""Pins the oracle (and the host-side sampling logic) to outputs of the
reference's own code, captured by tests/golden/make_golden.py (which ran
/root/reference/magma/{adapters,sampling,utils}.py in place).  CPU only."""
import os
import types

import pytest

import torch

from oracle import model as O

PINS = torch.load(os.path.join(os.path.dirname(__file__), "golden", "reference_pins.pt"), weights_only=False)


def test_adapter_matches_reference():
    pin = PINS["adapter"]
    p = {"a." + k.replace("adapter.", ""): v for k, v in pin["sd"].items()}
    y = O.adapter_fwd(p, "a.", pin["x"])
    assert torch.allclose(y, pin["y"], atol=1e-6, rtol=1e-6)
    assert PINS["adapter_init_absmax"] <= 2e-3 + 1e-9      # reference adapters.py:28-33 clamp
    q = O.init_params(O.OracleConfig.tiny(), 0)
    assert float(q["lm.transformer.h.0.mlp.1.adapter.0.weight"].abs().max()) <= 2e-3 + 1e-9


def test_adapter_wrapper_matches_reference():
    pin = PINS["adapter_wrapper"]
    p = {"a." + k.replace("adapter.", ""): v for k, v in pin["sd"].items() if k.startswith("adapter.")}
    attn_out = pin["x"] * 0.5 + 1.0                     # the toy attention block used by make_golden.py
    assert torch.allclose(O.adapter_fwd(p, "a.", attn_out), pin["y"], atol=1e-6, rtol=1e-6)
    assert pin["rest"] == ["present", "weights"]        # tuple tail is passed through untouched


def test_sampling_filters_match_reference():
    from magma_amd import sampling as S
    for mod in (O, S):
        for thr, key in ((0.9, "out_0.9"), (0.5, "out_0.5")):
            got = mod.top_p_filter(PINS["top_p"]["logits"].clone(), thr)
            assert torch.equal(got, PINS["top_p"][key]), (mod.__name__, thr)
        assert torch.equal(mod.top_k_filter(PINS["top_k"]["logits"].clone(), 5), PINS["top_k"]["out_5"])
    assert S.remove_tokens_after_eos(PINS["remove_eos"]["in"].clone(), 1, 7) == PINS["remove_eos"]["out"]
    assert S.remove_tokens_after_eos(PINS["remove_eos_none"]["in"].clone(), 1, 7) == PINS["remove_eos_none"]["out"]


def test_build_labels_matches_reference():
    pin = PINS["build_labels"]
    got = O.build_labels(pin["P"], pin["captions"], pin["eos"])
    assert torch.equal(got, pin["labels"])


def test_generate_loop_matches_reference():
    """Our sampling.generate driven by the same toy LM must emit the same token
    ids, strings and (prefill, then one-id-per-step) call pattern."""
    from magma_amd import sampling as S
    pin = PINS["generate_toy"]

    class ToyLM:
        def __init__(self, V):
            self.V, self.calls = V, []

        def __call__(self, inputs_embeds=None, input_ids=None, use_cache=None, past_key_values=None, cache_hint=None,
                     reuse_cache=False):
            from magma_amd.language_model import LMOutput
            if inputs_embeds is not None:
                last = (inputs_embeds[:, -1, :].sum(-1) * 7).long() % self.V
                seen = inputs_embeds.shape[1]
            else:
                last, seen = input_ids[:, -1], past_key_values + 1
            self.calls.append(("embeds" if inputs_embeds is not None else "ids", seen))
            idx = torch.arange(self.V)[None, :]
            logits = (-((idx - (last[:, None] * 3 + seen) % self.V) ** 2).float())[:, None, :]
            return LMOutput(logits=logits, past_key_values=seen, next_token=logits[:, -1].argmax(-1))

    class ToyModel:
        training = False

        def __init__(self):
            self.lm = ToyLM(17)
            self.eos_token, self.image_token = 16, 15
            self.device = torch.device("cpu")
            self.tokenizer = types.SimpleNamespace(decode=lambda ids: " ".join(map(str, ids)))

        def eval(self):
            return self

        def train(self, mode=True):
            return self

    tm = ToyModel()
    toks = S.generate(tm, pin["emb"], max_steps=6, temperature=0.0, decode=False)
    assert torch.equal(toks, pin["tokens"])
    assert tm.lm.calls == pin["calls"]
    assert S.generate(ToyModel(), pin["emb"], max_steps=6, temperature=0.0, decode=True) == pin["strings"]


# ---- preprocessing: the third-party arithmetic behind reference transforms.py:121-134 is Pillow's resampler ----
PRE_GEOMS = [(480, 640, 384), (300, 200, 224), (224, 224, 384), (1000, 750, 384), (97, 131, 384), (384, 384, 384),
             (500, 333, 224), (64, 64, 224)]


@pytest.mark.parametrize("H,W,n_px", PRE_GEOMS)
def test_preprocess_matches_pil(H, W, n_px):
    """oracle/preprocess.py (numpy integer restatement of Pillow's ImagingResample) against PIL itself, bit for bit,
    then the whole clip transform against the host pipeline that mirrors the reference (PIL + torch)."""
    import numpy as np
    import PIL.Image as PilImage
    from oracle.preprocess import clip_preprocess_u8, resize_bicubic_u8
    from magma_amd.transforms import clip_preprocess
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    nw, nh = (n_px, int(n_px * H / W)) if W <= H else (int(n_px * W / H), n_px)
    ref = np.asarray(PilImage.fromarray(img).resize((nw, nh), PilImage.BICUBIC))
    assert np.array_equal(resize_bicubic_u8(img, nw, nh), ref)
    host = clip_preprocess(n_px)(PilImage.fromarray(img))[0].numpy()
    assert np.array_equal(clip_preprocess_u8(img, n_px), host)


def test_product_coefficient_tables_equal_oracle():
    """the host half of the device path (magma_amd.transforms.pil_bicubic_tables) builds the same integer tables"""
    import numpy as np
    from oracle.preprocess import precompute_coeffs
    from magma_amd.transforms import pil_bicubic_tables
    for a, b in [(640, 512), (200, 224), (750, 288), (131, 384), (1000, 384), (333, 224), (64, 224), (4000, 384)]:
        kk, bounds = pil_bicubic_tables(a, b)
        rk, rb = precompute_coeffs(a, b)
        assert np.array_equal(kk.astype(np.int64), rk) and np.array_equal(bounds.astype(np.int64), rb)


def test_parallel_adapters_match_reference():
    """ParallelAdapter / ParallelAdapterWrapper (reference magma/adapters.py:42-92), run in place by make_golden.py
    around a toy wrapped module: y = module(x) + adapter(x) * adapter_scale."""
    for key, module_out in (("parallel_adapter", lambda x: torch.tanh(x) * 2.0 - 0.25),
                            ("parallel_adapter_scaled", lambda x: torch.tanh(x) * 2.0 - 0.25),
                            ("parallel_adapter_wrapper", lambda x: x * 0.5 + 1.0)):
        pin = PINS[key]
        p = {"a." + k.replace("adapter.", ""): v for k, v in pin["sd"].items() if k.startswith("adapter.")}
        if "adapter_scale" in pin["sd"]:
            p["scale"] = pin["sd"]["adapter_scale"]
        y = O.parallel_adapter_fwd(p, "a.", "scale", pin["x"], module_out(pin["x"]))
        assert torch.allclose(y, pin["y"], atol=1e-6, rtol=1e-6), key
    assert "adapter_scale" not in PINS["parallel_adapter"]["sd"]          # plain "parallel": the scale is the constant 1
    assert float(PINS["parallel_adapter_scaled"]["sd"]["adapter_scale"]) == 1.75
    assert PINS["parallel_adapter_wrapper"]["rest"] == ["present", "weights"]


@pytest.mark.parametrize("name,act", [("gelu", "gelu"), ("gelu_tanh", "gelu_tanh"), ("ln", "relu"), ("ln_gelu", "gelu")])
def test_adapter_options_match_reference(name, act):
    """reference adapters.py:11-24: ``activation`` and ``add_layernorm`` -- the reference's own Adapter run in place with each
    option (tests/golden/make_golden.py), against the oracle's statement of it; and the product module builds the same
    state-dict keys for every option set."""
    import functools
    pin = PINS["adapter_options"][name]
    p = {"a." + k.replace("adapter.", ""): v for k, v in pin["sd"].items()}
    y = O.adapter_fwd(p, "a.", pin["x"], act)
    assert torch.allclose(y, pin["y"], atol=2e-6, rtol=1e-5)
    from magma_amd.adapters import Adapter
    kw = {"gelu": dict(activation=torch.nn.GELU), "gelu_tanh": dict(activation=functools.partial(torch.nn.GELU, approximate="tanh")),
          "ln": dict(add_layernorm=True), "ln_gelu": dict(add_layernorm=True, activation=torch.nn.GELU)}[name]
    mod = Adapter(dim=64, downsample_factor=4, **kw)
    assert set(mod.state_dict()) == set(pin["sd"])
    assert (mod.ln is not None) == name.startswith("ln") and mod.down.weight.shape == (16, 64) and mod.up.weight.shape == (64, 16)
    assert not mod.plain
    with pytest.raises(NotImplementedError):
        Adapter(dim=64, activation=torch.nn.Tanh)
