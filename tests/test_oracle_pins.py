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


@pytest.mark.skipif(not os.path.isdir("/root/reference/magma"), reason="the reference tree is only present in the build container")
def test_oracle_forward_and_embed_plumbing_equal_the_reference_methods_run_in_place(monkeypatch):
    """reference magma/magma.py:195-212 (Magma.embed) and :238-276 (Magma.forward): the two method bodies are extracted with ast and
    executed IN PLACE on a stand-in `self` whose image_prefix / word_embedding are the oracle's pieces and whose `lm` records what it
    is handed -- together with the reference's own build_labels (utils.py:334-364).  What the reference hands its LM (inputs_embeds:
    prefix, then the caption embeddings cut so that the total is seq_len; labels) is exactly what oracle.magma_forward hands
    oracle.lm_forward, and oracle.embed equals the reference's embed on a [image, tokens, image] list.  This pins the PLUMBING of
    a3 / a11 (concat order, truncation, label construction); the arithmetic of the pieces is pinned elsewhere."""
    import ast
    from types import SimpleNamespace
    from typing import List, Optional
    import torch.nn.functional as F
    import oracle.model as om

    class TensorType:
        def __class_getitem__(cls, item):
            return cls

    ns = {"torch": torch, "TensorType": TensorType, "List": List, "Optional": Optional, "ModelOutput": object}
    for node in ast.parse(open("/root/reference/magma/utils.py").read()).body:
        if isinstance(node, ast.FunctionDef) and node.name == "build_labels":
            exec(compile(ast.Module(body=[node], type_ignores=[]), "/root/reference/magma/utils.py", "exec"), ns)
    for node in ast.parse(open("/root/reference/magma/magma.py").read()).body:
        if isinstance(node, ast.ClassDef) and node.name == "Magma":
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name in ("embed", "forward"):
                    item.decorator_list = []
                    exec(compile(ast.Module(body=[item], type_ignores=[]), "/root/reference/magma/magma.py", "exec"), ns)
    cfg = om.OracleConfig.tiny(n_positions=64)
    p = om.init_params(cfg, seed=5)
    g = torch.Generator().manual_seed(11)
    B = 2
    images = torch.randn(B, 3, 64, 64, generator=g)
    caps = torch.full((B, 64), cfg.eos_token, dtype=torch.int64)
    caps[0, :13] = torch.randint(0, 1000, (13,), generator=g)
    caps[1, :5] = torch.randint(0, 1000, (5,), generator=g)
    handed = {}

    def lm(inputs_embeds=None, labels=None, output_hidden_states=False):
        handed["ref"] = (inputs_embeds.clone(), labels.clone())
        return "lm-output"

    fake = SimpleNamespace(image_prefix=lambda x: om.image_prefix_fwd(p, cfg, x.float()),
                           word_embedding=lambda ids: F.embedding(ids, p["lm.transformer.wte.weight"]),
                           lm=lm, seq_len=64, eos_token=cfg.eos_token, device=torch.device("cpu"))
    assert ns["forward"](fake, images=images, captions=caps) == "lm-output"

    def record(p_, cfg_, inputs_embeds=None, labels=None, **kw):
        handed["oracle"] = (inputs_embeds.clone(), labels.clone())
        return {"loss": torch.zeros(()), "logits": None}
    monkeypatch.setattr(om, "lm_forward", record)
    om.magma_forward(p, cfg, images, caps)
    assert torch.equal(handed["ref"][1], handed["oracle"][1])                      # labels: integer, exact
    assert handed["ref"][0].shape == handed["oracle"][0].shape == (B, 64, cfg.d_model)
    assert torch.equal(handed["ref"][0], handed["oracle"][0])
    with pytest.raises(AssertionError):
        ns["forward"](fake, images=images, captions=caps[:, :60])                  # captions must be padded to seq_len
    # embed: [image, tokens, image] (the reference casts images with .half(): the stand-in prefix takes them back to fp32)
    toks = torch.randint(0, 1000, (B, 7), generator=g)
    ref_emb = ns["embed"](fake, [images, toks, images])
    assert torch.allclose(ref_emb, om.embed(p, cfg, [images.half().float(), toks, images.half().float()]), atol=0, rtol=0)
    with pytest.raises(ValueError):
        ns["embed"](fake, [torch.zeros(3)])


@pytest.mark.skipif(not os.path.isdir("/root/reference/magma"), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("adapter_type", ["normal", "parallel", "scaled_parallel"])
@pytest.mark.parametrize("kwargs", [{}, {"add_layernorm": True}, {"downsample_factor": 8}])
def test_add_adapters_builds_the_reference_module_tree(adapter_type, kwargs):
    """reference magma/magma.py:102-174 (Magma.add_adapters), the method body extracted with ast and executed IN PLACE with the reference's
    own adapter classes (magma/adapters.py imported in place) on a toy `self` (ModuleList of blocks with .attn / .mlp), next to this
    repo's Magma.add_adapters on an identical toy: the same parameter names -- the checkpoint keys of SURVEY Q8 -- with the same shapes
    for every adapter type at both locations and with the reference's options; a second call at the same location raises on both."""
    import ast
    import importlib.util
    import sys
    from types import SimpleNamespace
    from typing import Literal
    import torch.nn as nn
    from magma_amd.magma import Magma
    tt = types.ModuleType("torchtyping")

    class TensorType:
        def __class_getitem__(cls, item):
            return cls
    tt.TensorType, tt.patch_typeguard = TensorType, (lambda: None)
    sys.modules.setdefault("torchtyping", tt)
    spec = importlib.util.spec_from_file_location("ref_adapters_tree", "/root/reference/magma/adapters.py")
    ra = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ra)
    ns = {"nn": nn, "Literal": Literal, "Adapter": ra.Adapter, "ParallelAdapter": ra.ParallelAdapter,
          "AdapterWrapper": ra.AdapterWrapper, "ParallelAdapterWrapper": ra.ParallelAdapterWrapper}
    for node in ast.parse(open("/root/reference/magma/magma.py").read()).body:
        if isinstance(node, ast.ClassDef) and node.name == "Magma":
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == "add_adapters":
                    exec(compile(ast.Module(body=[item], type_ignores=[]), "/root/reference/magma/magma.py", "exec"), ns)

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.attn = nn.Linear(16, 16, bias=False)
            self.mlp = nn.Sequential(nn.Linear(16, 64), nn.Linear(64, 16))

    def toy():
        tr = nn.ModuleList([Block() for _ in range(3)])
        return SimpleNamespace(transformer=tr, mlp_adapter_added=False, attn_adapter_added=False, device=torch.device("cpu"),
                               dtype=torch.float32, lm=SimpleNamespace(config=SimpleNamespace(hidden_size=16), invalidate_packed=lambda: None))
    theirs, mine = toy(), toy()
    for location in ("mlp", "attention"):
        ns["add_adapters"](theirs, adapter_type=adapter_type, location=location, **kwargs)
        Magma.add_adapters(mine, adapter_type=adapter_type, location=location, **kwargs)
    kt = {n: tuple(p.shape) for n, p in theirs.transformer.named_parameters()}
    km = {n: tuple(p.shape) for n, p in mine.transformer.named_parameters()}
    assert kt == km and len(kt) > 3 * 3
    assert theirs.mlp_adapter_added and mine.mlp_adapter_added and theirs.attn_adapter_added and mine.attn_adapter_added
    for obj, fn in ((theirs, ns["add_adapters"]), (mine, Magma.add_adapters)):
        with pytest.raises(ValueError):
            fn(obj, adapter_type=adapter_type, location="mlp")


@pytest.mark.skipif(not os.path.isdir("/root/reference/magma"), reason="the reference tree is only present in the build container")
def test_preprocess_inputs_equals_the_reference_method_run_in_place(tmp_path):
    """reference magma/magma.py:176-193 (Magma.preprocess_inputs) with the reference's own ImageInput (magma/image_input.py:6-23), both
    executed IN PLACE, next to this repo's method and ImageInput on the same stand-in `self` (same tokenizer, same transform, an embed
    that records): the caller's list is mutated to the same tensors (token ids (1, T) int64, image (1, 3, n, n)), embed is handed the same
    list, embed=False returns the list itself, an unknown item type raises."""
    import ast
    import importlib.util
    from types import SimpleNamespace
    from typing import List
    import numpy as np
    import PIL.Image as I
    from magma_amd.image_input import ImageInput
    from magma_amd.magma import Magma
    from magma_amd.tokenizer import ByteTokenizer
    from magma_amd.transforms import clip_preprocess
    spec = importlib.util.spec_from_file_location("ref_image_input", "/root/reference/magma/image_input.py")
    rii = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rii)
    ns = {"torch": torch, "List": List, "ImageInput": rii.ImageInput}
    for node in ast.parse(open("/root/reference/magma/magma.py").read()).body:
        if isinstance(node, ast.ClassDef) and node.name == "Magma":
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == "preprocess_inputs":
                    exec(compile(ast.Module(body=[item], type_ignores=[]), "/root/reference/magma/magma.py", "exec"), ns)
    path = str(tmp_path / "img.png")
    I.fromarray((np.random.RandomState(0).rand(45, 70, 3) * 255).astype("uint8")).save(path)
    tok, tf = ByteTokenizer(64), clip_preprocess(32)
    seen = {}

    def fake(tag):
        def embed(lst):
            seen[tag] = [t.clone() for t in lst]
            return "embedded"
        return SimpleNamespace(tokenizer=tok, transforms=tf, embed=embed)
    l_ref = [rii.ImageInput(path), "Describe the painting:", rii.ImageInput(path)]
    l_mine = [ImageInput(path), "Describe the painting:", ImageInput(path)]
    assert ns["preprocess_inputs"](fake("ref"), l_ref) == "embedded" and Magma.preprocess_inputs(fake("mine"), l_mine) == "embedded"
    assert len(l_ref) == len(l_mine) == 3
    for a, b, c, d in zip(l_ref, l_mine, seen["ref"], seen["mine"]):
        assert torch.is_tensor(a) and a.dtype == b.dtype and torch.equal(a, b) and torch.equal(c, d) and torch.equal(a, c)
    assert l_mine[0].shape == (1, 3, 32, 32) and l_mine[1].dtype == torch.int64 and l_mine[1].shape[0] == 1
    keep_ref, keep_mine = ["x"], ["x"]
    assert ns["preprocess_inputs"](fake("r"), keep_ref, embed=False) is keep_ref and Magma.preprocess_inputs(fake("m"), keep_mine, embed=False) is keep_mine
    for fn, who in ((ns["preprocess_inputs"], "ref"), (Magma.preprocess_inputs, "mine")):
        with pytest.raises(Exception, match="Invalid input type"):
            fn(fake(who), [3.14])


@pytest.mark.skipif(not os.path.isdir("/root/reference/magma"), reason="the reference tree is only present in the build container")
def test_oracle_image_prefix_plumbing_equals_the_reference_method_run_in_place():
    """reference magma/image_prefix.py:78-109 (ImagePrefix.forward) extracted with ast and executed IN PLACE (with the file's own
    ENCODER_SEQ_LENS table and einops.rearrange) on a stand-in `self` whose encoder / Linear / LayerNorm carry the oracle's tensors:
    (a) sequence encoders (clip_resnet_large: features (B, HW, C), no reshape) == oracle.image_prefix_fwd; (b) pooled encoders
    (features (B, C, 1, 1) -> Linear to seq*d -> "b (s d) -> b s d") == oracle.pooled_prefix_fwd."""
    import ast
    from types import SimpleNamespace
    import torch.nn as nn
    from einops import rearrange
    import oracle.model as om
    src = open("/root/reference/magma/image_prefix.py").read()
    ns = {"torch": torch, "nn": nn, "rearrange": rearrange, "TensorType": type("T", (), {"__class_getitem__": classmethod(lambda c, i: c)})}
    for node in ast.parse(src).body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") in ("ENCODER_SEQ_LENS", "ENCODER_OUT_DIMS"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), "/root/reference/magma/image_prefix.py", "exec"), ns)
        if isinstance(node, ast.ClassDef) and node.name == "ImagePrefix":
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == "forward":
                    exec(compile(ast.Module(body=[item], type_ignores=[]), "/root/reference/magma/image_prefix.py", "exec"), ns)
    assert ns["ENCODER_SEQ_LENS"]["clip_resnet_large"] == 144
    cfg = om.OracleConfig.tiny()
    p = om.init_params(cfg, seed=9)
    images = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(2))

    def linear(w, b):
        m = nn.Linear(w.shape[1], w.shape[0])
        m.weight.data, m.bias.data = w.clone(), b.clone()
        return m
    ln = nn.LayerNorm(cfg.d_model, eps=cfg.ln_eps)
    ln.weight.data, ln.bias.data = p["image_prefix.ln.weight"].clone(), p["image_prefix.ln.bias"].clone()
    fake = SimpleNamespace(enc=lambda x: om.encoder_fwd(p, cfg, x), encoder_type="clip_resnet_large", dropout=nn.Identity(),
                           proj=linear(p["image_prefix.proj.weight"], p["image_prefix.proj.bias"]), use_layernorm=cfg.use_prefix_ln, ln=ln,
                           out_dim=cfg.d_model, out_seq_len=None)
    with torch.no_grad():
        got, want = ns["forward"](fake, images), om.image_prefix_fwd(p, cfg, images)
    assert got.shape == want.shape and got.ndim == 3 and torch.allclose(got, want, atol=1e-6, rtol=1e-6)
    # pooled encoder: (B, C, 1, 1) features, out_seq_len tokens from one Linear
    C, S, d = 24, 3, 16
    g = torch.Generator().manual_seed(4)
    pp = {"image_prefix.proj.weight": torch.randn(S * d, C, generator=g) * 0.1, "image_prefix.proj.bias": torch.randn(S * d, generator=g) * 0.1,
          "image_prefix.ln.weight": torch.rand(d, generator=g) + 0.5, "image_prefix.ln.bias": torch.randn(d, generator=g) * 0.1}
    feats = torch.randn(2, C, generator=g)
    ln2 = nn.LayerNorm(d)
    ln2.weight.data, ln2.bias.data = pp["image_prefix.ln.weight"].clone(), pp["image_prefix.ln.bias"].clone()
    fake2 = SimpleNamespace(enc=lambda x: feats[:, :, None, None], encoder_type="nfresnet50", dropout=nn.Identity(),
                            proj=linear(pp["image_prefix.proj.weight"], pp["image_prefix.proj.bias"]), use_layernorm=True, ln=ln2, out_dim=d, out_seq_len=S)
    with torch.no_grad():
        got2, want2 = ns["forward"](fake2, images), om.pooled_prefix_fwd(pp, d, S, feats)
    assert got2.shape == (2, S, d) and torch.allclose(got2, want2, atol=1e-6, rtol=1e-6)


def test_oracle_window_is_exact():
    """tests/fullwidth_common.oracle_window: the S = 2048 GPU parity tests evaluate the CPU oracle on the first
    ceil64(P + longest caption + 2) positions only.  Loss, labels, target-row logits and every gradient must be the values of the
    full-length evaluation (causal mask: no position influences an earlier one; build_labels masks everything behind the first
    eos, reference magma/utils.py:334-364) -- checked here on the tiny configuration, full length 256 against the window."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fullwidth_common as F
    cfg = O.OracleConfig.tiny(n_positions=256)
    params = O.init_params(cfg, seed=3)
    for k in params:
        if ".adapter." in k:
            params[k] = params[k] * 20
    g = torch.Generator().manual_seed(2)
    B, S, P = 2, 256, 4
    images = torch.randn(B, 3, 64, 64, generator=g)
    caps = torch.full((B, S), cfg.eos_token, dtype=torch.int64)
    caps[0, :37] = torch.randint(0, 1000, (37,), generator=g)
    caps[1, :9] = torch.randint(0, 1000, (9,), generator=g)
    mask = (torch.rand(B, P, cfg.d_model, generator=g) < 0.9).float() / 0.9
    win = F.oracle_window(caps, P, cfg.eos_token)
    assert win.shape == (B, 64) and torch.equal(win, caps[:, :64])
    names = [k for k in params if (".adapter." in k or k.startswith("image_prefix.")) and "running_" not in k]

    def run(c):
        p = {k: (v.detach().clone() if v.is_floating_point() else v) for k, v in params.items()}
        for k in names:
            p[k].requires_grad_(True)
        out = O.magma_forward(p, cfg, images, c, dropout_mask=mask)
        out["loss"].backward()
        return out, {k: p[k].grad for k in names}

    full, g_full = run(caps)
    part, g_part = run(win)
    assert torch.equal(full["labels"][:, :64], part["labels"]) and bool((full["labels"][:, 64:] == -100).all())
    assert abs(float(full["loss"]) - float(part["loss"])) <= 1e-6 * abs(float(full["loss"]))
    rows = (part["labels"][0, 1:] != -100).nonzero().squeeze(1)
    assert torch.allclose(full["logits"][0, rows], part["logits"][0, rows], rtol=1e-5, atol=1e-5)
    for k in names:
        a, b = g_full[k], g_part[k]
        assert float((a - b).norm()) <= 1e-5 * float(a.norm()) + 1e-12, k
