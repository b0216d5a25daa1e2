"""Generates tests/golden/reference_pins.pt by RUNNING the reference's own code
from /root/reference (read-only) in this container -- nothing is copied:

  * magma/adapters.py and magma/sampling.py are imported as-is (a 6-line
    annotation-only ``torchtyping`` shim is put in sys.modules first);
  * magma/utils.py cannot be imported (deepspeed / wandb / gdown absent), so the
    ``build_labels`` FunctionDef is extracted with ``ast`` and executed in place.

The fixtures pin the oracle (oracle/model.py) and the host-side sampling logic
(magma_amd/sampling.py) to the reference.  Re-run:  python tests/golden/make_golden.py
"""
import ast
import importlib.util
import os
import sys
import types

import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_pins.pt")


def _shim_torchtyping():
    m = types.ModuleType("torchtyping")

    class TensorType:
        def __class_getitem__(cls, item):
            return cls

    m.TensorType = TensorType
    m.patch_typeguard = lambda: None
    sys.modules["torchtyping"] = m


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _extract_function(rel, fname, namespace):
    src = open(os.path.join(REF, rel)).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == fname:
            code = compile(ast.Module(body=[node], type_ignores=[]), os.path.join(REF, rel), "exec")
            exec(code, namespace)
            return namespace[fname]
    raise KeyError(fname)


def main():
    _shim_torchtyping()
    adapters = _load("ref_adapters", "magma/adapters.py")
    sampling = _load("ref_sampling", "magma/sampling.py")
    from torchtyping import TensorType
    build_labels = _extract_function("magma/utils.py", "build_labels", {"torch": torch, "TensorType": TensorType})
    pins = {}

    # ---- Adapter / AdapterWrapper (reference magma/adapters.py:6-39,95-116) ----
    torch.manual_seed(0)
    ad = adapters.Adapter(dim=64, downsample_factor=4)
    x = torch.randn(3, 5, 64)
    pins["adapter"] = {"sd": {k: v.clone() for k, v in ad.state_dict().items()}, "x": x, "y": ad(x).detach()}
    pins["adapter_init_absmax"] = float(max(p.abs().max() for p in ad.parameters()))

    class FakeAttn(torch.nn.Module):
        def forward(self, x, *a, **k):
            return (x * 0.5 + 1.0, "present", "weights")

    aw = adapters.AdapterWrapper(attn_block=FakeAttn(), dim=64, downsample_factor=8)
    out = aw(x)
    pins["adapter_wrapper"] = {"sd": {k: v.clone() for k, v in aw.state_dict().items()}, "x": x,
                               "y": out[0].detach(), "rest": list(out[1:])}

    # ---- ParallelAdapter / ParallelAdapterWrapper (reference magma/adapters.py:42-92) ----
    class FakeMlp(torch.nn.Module):
        def forward(self, x):
            return torch.tanh(x) * 2.0 - 0.25

    for scaled in (False, True):
        pa = adapters.ParallelAdapter(module=FakeMlp(), dim=64, downsample_factor=4, scaled=scaled)
        if scaled:
            with torch.no_grad():
                pa.adapter_scale.fill_(1.75)
        pins["parallel_adapter_scaled" if scaled else "parallel_adapter"] = {
            "sd": {k: v.clone() for k, v in pa.state_dict().items()}, "x": x, "y": pa(x).detach()}
    paw = adapters.ParallelAdapterWrapper(module=FakeAttn(), dim=64, downsample_factor=8, scaled=True)
    with torch.no_grad():
        paw.adapter_scale.fill_(0.6)
    out = paw(x)
    pins["parallel_adapter_wrapper"] = {"sd": {k: v.clone() for k, v in paw.state_dict().items()}, "x": x,
                                        "y": out[0].detach(), "rest": list(out[1:])}

    # ---- sampling filters (reference magma/sampling.py:7-40) ----
    g = torch.Generator().manual_seed(1)
    logits = torch.randn(4, 50, generator=g) * 3
    logits[1] = torch.randn(50, generator=g) * 0.1           # flat row: top-1 prob < 0.1 -> filter active
    pins["top_p"] = {"logits": logits, "out_0.9": sampling.top_p_filter(logits.clone(), 0.9),
                     "out_0.5": sampling.top_p_filter(logits.clone(), 0.5)}
    pins["top_k"] = {"logits": logits, "out_5": sampling.top_k_filter(logits.clone(), 5)}
    t = torch.tensor([7, 7, 3, 9, 1, 4, 1, 5])
    pins["remove_eos"] = {"in": t, "out": sampling.remove_tokens_after_eos(t.clone(), 1, 7)}
    t2 = torch.tensor([7, 3, 9, 4])
    pins["remove_eos_none"] = {"in": t2, "out": sampling.remove_tokens_after_eos(t2.clone(), 1, 7)}

    # ---- build_labels (reference magma/utils.py:334-364) ----
    eos, S, P = 11, 24, 5
    g = torch.Generator().manual_seed(2)
    cap = torch.randint(12, 40, (6, S), generator=g)
    cap[0, 9:] = eos
    cap[1, 0] = eos
    cap[2, :] = eos
    cap[4, S - P + 1] = eos         # eos only in the truncated tail
    cap[5, 3] = eos
    cap[5, 10] = eos
    emb = torch.zeros(6, P, 8)
    pins["build_labels"] = {"captions": cap, "P": P, "eos": eos,
                            "labels": build_labels(emb, cap.clone(), eos, torch.device("cpu"))}

    # ---- generate() loop semantics with a deterministic toy LM (sampling.py:43-121) ----
    class Out:
        pass

    class ToyLM:
        """logits depend on (last token, #tokens seen) so cache misuse shows up."""

        def __init__(self, V):
            self.V = V
            self.calls = []

        def __call__(self, inputs_embeds=None, input_ids=None, use_cache=None, past_key_values=None):
            o = Out()
            if inputs_embeds is not None:
                b, s, _ = inputs_embeds.shape
                last = (inputs_embeds[:, -1, :].sum(-1) * 7).long() % self.V
                seen = s
            else:
                b = input_ids.shape[0]
                last = input_ids[:, -1]
                seen = past_key_values + 1
            self.calls.append(("embeds" if inputs_embeds is not None else "ids", seen))
            idx = torch.arange(self.V)[None, :]
            o.logits = (-((idx - (last[:, None] * 3 + seen) % self.V) ** 2).float())[:, None, :]
            o.past_key_values = seen
            return o

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

    emb = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(2, 3, 4) / 10
    tm = ToyModel()
    toks = sampling.generate(tm, emb, max_steps=6, temperature=0.0, decode=False)
    tm2 = ToyModel()
    strs = sampling.generate(tm2, emb, max_steps=6, temperature=0.0, decode=True)
    pins["generate_toy"] = {"emb": emb, "tokens": toks, "strings": strs, "calls": tm.lm.calls}

    # ---- Adapter options (reference magma/adapters.py:11-24): activation, add_layernorm -- appended last, own seed, so that every
    #      entry above keeps its random stream ----
    import functools
    torch.manual_seed(4321)
    xo = torch.randn(3, 5, 64)
    opts = {}
    for name, kw in (("gelu", dict(activation=torch.nn.GELU)),
                     ("gelu_tanh", dict(activation=functools.partial(torch.nn.GELU, approximate="tanh"))),
                     ("ln", dict(add_layernorm=True)),
                     ("ln_gelu", dict(add_layernorm=True, activation=torch.nn.GELU))):
        ad = adapters.Adapter(dim=64, downsample_factor=4, **kw)
        with torch.no_grad():                       # weights large enough for the activation's shape to matter; LayerNorm off identity
            for prm in ad.parameters():
                prm.mul_(300.0) if prm.ndim == 2 else None
            for m in ad.modules():
                if isinstance(m, torch.nn.LayerNorm):
                    m.weight.add_(torch.randn(64) * 0.1)
                    m.bias.add_(torch.randn(64) * 0.1)
        opts[name] = {"sd": {k: v.clone() for k, v in ad.state_dict().items()}, "x": xo, "y": ad(xo).detach()}
    pins["adapter_options"] = opts

    torch.save(pins, OUT)
    print("wrote", OUT, {k: type(v).__name__ for k, v in pins.items()})


if __name__ == "__main__":
    main()
