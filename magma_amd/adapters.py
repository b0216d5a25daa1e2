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


class Adapter(nn.Module):
    def __init__(self, dim: int, downsample_factor: int = 4, activation=nn.ReLU, add_layernorm: bool = False,
                 device=None, dtype=None):
        super().__init__()
        if add_layernorm or activation is not nn.ReLU:
            raise NotImplementedError("only ReLU adapters without LayerNorm are on the MAGMA_v1/v2 path (SURVEY Q12)")
        kw = dict(device=device, dtype=dtype)
        self.adapter = nn.Sequential(nn.Linear(dim, dim // downsample_factor, **kw), nn.ReLU(),
                                     nn.Linear(dim // downsample_factor, dim, **kw))
        for m in self.adapter:
            m._is_adapter = True   # GPTJForCausalLM.init_weights leaves adapters alone
        self.adapter.apply(self.init_weights)

    @staticmethod
    def init_weights(m: nn.Module, std=1e-3):
        if isinstance(m, nn.Linear):
            with torch.no_grad():
                m.weight.normal_(std=std).clamp_(-2 * std, 2 * std)
                m.bias.normal_(std=std).clamp_(-2 * std, 2 * std)

    def adapter_branch(self, x: torch.Tensor, residual: torch.Tensor = None) -> torch.Tensor:
        """W_up relu(W_dn x + b_dn) + b_up (+ residual) for x (..., dim) on the GPU: two MFMA GEMMs."""
        from . import ops
        dn, up = self.adapter[0], self.adapter[2]
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).to(torch.bfloat16).contiguous()
        with torch.cuda.device(x2.device):
            t = ops.gemm(x2, ops.RawWeight(ops.pad_k_rowmajor(dn.weight.detach().to(torch.bfloat16)), K=dn.weight.shape[1],
                                           bias=dn.bias.detach().float().contiguous()), act=ops.MG_ACT_RELU, layout="rm")
            res = () if residual is None else (residual.reshape(-1, shape[-1]).to(torch.bfloat16).contiguous(),)
            y = ops.gemm(t, ops.RawWeight(ops.pad_k_rowmajor(up.weight.detach().to(torch.bfloat16)), K=up.weight.shape[1],
                                          bias=up.bias.detach().float().contiguous()), residuals=res, layout="rm")
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
