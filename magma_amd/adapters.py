"""Adapter modules.  Module / parameter names follow reference magma/adapters.py
(``adapter.0`` = down Linear, ``adapter.2`` = up Linear, ``adapter_scale``, ``module``
/ ``attn_block`` for the wrapped block; init N(0,1e-3) clamped to +-2e-3) so
reference checkpoints load by name.

The arithmetic runs on the HIP kernels.  Inside the model graph the engines
(engine.py / train_engine.py) fuse it into the surrounding GEMM epilogues:

    normal           y = m + W_up relu(W_dn m + b_dn) + b_up          m = wrapped block's output   (adapters.py:38-39)
    parallel         y = m + s * (W_up relu(W_dn x + b_dn) + b_up)    x = wrapped block's INPUT    (adapters.py:62-65)
    scaled_parallel  the same with s = the trainable ``adapter_scale``; plain ``parallel`` has s = 1

``Adapter.forward`` / ``ParallelAdapter.adapter_branch`` run the same two GEMMs
standalone (bias + ReLU and bias + residual in the epilogues) for callers that
use an adapter as a module of its own."""
import torch
import torch.nn as nn


# Raw-pointer writers of adapter parameters (the engines' fused AdamW, checkpoint loads through the flat groups) do not bump a
# tensor's _version and keep its data_ptr: they bump THIS counter instead, and Adapter.adapter_branch keys its packed operands
# on it next to (_version, data_ptr) of weight and bias -- stale packs are dropped without repacking on every call.
_WEIGHTS_EPOCH = 0


def bump_weights_epoch() -> int:
    global _WEIGHTS_EPOCH
    _WEIGHTS_EPOCH += 1
    return _WEIGHTS_EPOCH


def activation_codes(act: nn.Module):
    """(epilogue activation, epilogue gradient mode, gradient needs the PRE-activation) of an adapter activation module.
    reference adapters.py:11,20 takes any nn.Module class; the fused epilogues cover ReLU (the default), torch.nn.GELU() (erf) and
    its tanh form (= HF's gelu_new); anything else has no kernel and is refused loudly."""
    from . import ops
    if isinstance(act, nn.ReLU):
        return ops.MG_ACT_RELU, ops.MG_AUX_RELU_GATE, False          # gate from the OUTPUT (t > 0), nothing else to keep
    if isinstance(act, nn.GELU):
        if getattr(act, "approximate", "none") == "tanh":
            return ops.MG_ACT_GELU_NEW, ops.MG_AUX_GELU_GRAD, True
        return ops.MG_ACT_GELU_ERF, ops.MG_AUX_GELU_ERF_GRAD, True
    raise NotImplementedError(f"adapter activation {type(act).__name__}: the fused epilogues cover nn.ReLU and nn.GELU (erf / tanh)")


class Adapter(nn.Module):
    """reference adapters.py:6-39.  ``adapter`` = Sequential([LayerNorm(dim)] if add_layernorm, Linear(dim, dim // f), activation(),
    Linear(dim // f, dim)) -- same module indices, hence the same checkpoint keys, as the reference for every option."""

    def __init__(self, dim: int, downsample_factor: int = 4, activation=nn.ReLU, add_layernorm: bool = False,
                 device=None, dtype=None):
        super().__init__()
        kw = dict(device=device, dtype=dtype)
        act = activation()
        activation_codes(act)                     # refuse what no kernel computes, at construction
        layers = [nn.LayerNorm(dim, **kw)] if add_layernorm else []
        layers += [nn.Linear(dim, dim // downsample_factor, **kw), act, nn.Linear(dim // downsample_factor, dim, **kw)]
        self.adapter = nn.Sequential(*layers)
        for m in self.adapter:
            m._is_adapter = True   # GPTJForCausalLM.init_weights leaves adapters alone
        self.adapter.apply(self.init_weights)

    # the three pieces, whatever the option set (the engines never index the Sequential themselves)
    @property
    def ln(self):
        return self.adapter[0] if isinstance(self.adapter[0], nn.LayerNorm) else None

    @property
    def down(self) -> nn.Linear:
        return self.adapter[1] if isinstance(self.adapter[0], nn.LayerNorm) else self.adapter[0]

    @property
    def up(self) -> nn.Linear:
        return self.adapter[-1]

    @property
    def act(self) -> nn.Module:
        return self.adapter[-2]

    @property
    def plain(self) -> bool:
        """ReLU bottleneck without LayerNorm: the shape every fused / folded launch structure is written for."""
        return self.ln is None and isinstance(self.act, nn.ReLU)

    @staticmethod
    def init_weights(m: nn.Module, std=1e-3):
        if isinstance(m, nn.Linear):
            with torch.no_grad():
                m.weight.normal_(std=std).clamp_(-2 * std, 2 * std)
                m.bias.normal_(std=std).clamp_(-2 * std, 2 * std)
        elif isinstance(m, nn.LayerNorm):
            with torch.no_grad():
                m.bias.zero_()
                m.weight.fill_(1.0)

    def adapter_branch(self, x: torch.Tensor, residual: torch.Tensor = None) -> torch.Tensor:
        """W_up act(W_dn [LN] x + b_dn) + b_up (+ residual) for x (..., dim) on the GPU: two MFMA GEMMs (+ a LayerNorm launch)."""
        from . import ops
        dn, up = self.down, self.up
        code, _, _ = activation_codes(self.act)
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).to(torch.bfloat16).contiguous()
        with torch.cuda.device(x2.device):
            xin = x2
            if self.ln is not None:
                xin = ops.layernorm(x2, self.ln.weight.detach().float().contiguous(), self.ln.bias.detach().float().contiguous(), self.ln.eps)
            # padded row-major operands of the live parameters, cached on (epoch, _version, data_ptr) of each weight AND bias:
            # in-place torch updates bump _version, a re-assigned .data changes data_ptr, and the engines' raw-pointer
            # optimizer step bumps the module-level epoch (bump_weights_epoch).  This is the module-call convenience path
            # (reference adapters.py:38-39); the engines keep their own packed operands.
            key = (_WEIGHTS_EPOCH,) + tuple((t._version, t.data_ptr(), t.device) for t in (dn.weight, dn.bias, up.weight, up.bias))
            cached = self.__dict__.get("_pack_cache")
            if cached is None or cached[0] != key:
                cached = (key,
                    ops.RawWeight(ops.pad_k_rowmajor(dn.weight.detach().to(torch.bfloat16)), K=dn.weight.shape[1],
                                  bias=dn.bias.detach().float().contiguous()),
                    ops.RawWeight(ops.pad_k_rowmajor(up.weight.detach().to(torch.bfloat16)), K=up.weight.shape[1],
                                  bias=up.bias.detach().float().contiguous()))
                self.__dict__["_pack_cache"] = cached
            packs = cached
            erf = code == ops.MG_ACT_GELU_ERF
            t = ops.gemm(xin, packs[1], act=ops.MG_ACT_NONE if erf else code, layout="rm")
            if erf:
                ops.gelu_erf(t, out=t)
            res = () if residual is None else (residual.reshape(-1, shape[-1]).to(torch.bfloat16).contiguous(),)
            y = ops.gemm(t, packs[2], residuals=res, layout="rm")
        return y.reshape(shape)

    def forward(self, x):
        """reference adapters.py:38-39: adapter(x) + x."""
        return self.adapter_branch(x, residual=x)


class AdapterWrapper(Adapter):
    """Attention + adapter (reference adapters.py:95-116): holds the wrapped
    attention parameters as ``attn_block`` and the adapter as ``adapter``."""

    def __init__(self, attn_block: nn.Module, dim: int, downsample_factor: int = 4, activation=nn.ReLU,
                 add_layernorm: bool = False, device=None, dtype=None):
        super().__init__(dim, downsample_factor, activation, add_layernorm, device=device, dtype=dtype)
        self.attn_block = attn_block


class ParallelAdapter(Adapter):
    """reference adapters.py:42-65: y = module(x) + adapter(x) * adapter_scale, the adapter reading the wrapped
    module's INPUT.  ``module`` keeps the reference's attribute name (checkpoint keys ``...mlp.module.c_fc...``)."""

    def __init__(self, module: nn.Module, dim: int, downsample_factor: int = 4, scaled: bool = False,
                 add_layernorm: bool = False, activation=nn.ReLU, device=None, dtype=None):
        super().__init__(dim, downsample_factor, activation, add_layernorm, device=device, dtype=dtype)
        self.module = module
        if scaled:
            self.adapter_scale = nn.Parameter(torch.ones(1, device=device, dtype=dtype))
        else:
            self.adapter_scale = 1

    def scale_value(self) -> float:
        s = self.adapter_scale
        return float(s.detach().float()) if torch.is_tensor(s) else float(s)

    def forward(self, x, **module_kwargs):
        raise RuntimeError("ParallelAdapter wraps a GPT-J sub-block whose arithmetic lives in the HIP engine; "
                           "call the model (magma_amd.engine) or adapter_branch(x) for the adapter alone")


class ParallelAdapterWrapper(ParallelAdapter):
    """reference adapters.py:68-92: the attention-side parallel adapter."""
