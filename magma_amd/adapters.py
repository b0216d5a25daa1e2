"""Adapter parameter containers.  Module / parameter names follow reference
magma/adapters.py (``adapter.0`` = down Linear, ``adapter.2`` = up Linear, init
N(0,1e-3) clamped to +-2e-3) so reference checkpoints load by name.  The
arithmetic  x + W_up relu(W_dn x + b_dn) + b_up  runs as two HIP GEMMs with
fused bias/ReLU/residual epilogues (engine.py); these classes hold parameters."""
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

    def forward(self, x):
        raise RuntimeError("Adapter is executed by the HIP engine (magma_amd.engine), not as a torch module")


class AdapterWrapper(Adapter):
    """Attention + adapter (reference adapters.py:95-116): holds the wrapped
    attention parameters as ``attn_block`` and the adapter as ``adapter``."""

    def __init__(self, attn_block: nn.Module, dim: int, downsample_factor: int = 4, activation=nn.ReLU,
                 add_layernorm: bool = False, device=None, dtype=None):
        super().__init__(dim, downsample_factor, activation, add_layernorm, device=device, dtype=dtype)
        self.attn_block = attn_block
