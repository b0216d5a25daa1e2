"""Shared by tests/test_fullwidth_gpu.py and tools/find_margin_seed.py: the full-WIDTH (d 4096, 16 heads, ff 16384,
V 50258, RN50x16 trunk), one-LAYER configuration of the parity tests at BASELINE shapes, its seeded weights and inputs.

One layer keeps the fp32 CPU oracle at seconds ("full-dim single block"); every kernel variant the
28-layer headline runs (K = 16384 GEMV, the fused 28 672-column ln_1+qkv+fc_in GEMV, decode attention co-launch,
split-K prefill at M = 456, the 50 258-column head) is inside the comparison."""
import torch

WEIGHT_SEED = 21
GREEDY_STEPS = 16
GREEDY_B = 2
PREFILL_LEN = 57          # 49 prefix tokens (224^2 image) + 8 prompt tokens: BASELINE config[1]
# input seed for the free-running greedy test, found by tools/find_margin_seed.py: with it EVERY one of the
# GREEDY_B x GREEDY_STEPS top-1 decisions of the fp32 oracle has a top-1/top-2 gap above SEARCH_MARGIN x std(logits)
# (margin-controlled inputs; the test itself demands TEST_MARGIN, a quarter of it)
GREEDY_INPUT_SEED = None  # filled in below
SEARCH_MARGIN = 0.06
TEST_MARGIN = 0.015


def full_width_config(n_positions: int = 2048, **kw):
    from oracle.model import OracleConfig
    base = dict(n_layer=1, n_positions=n_positions)
    base.update(kw)
    return OracleConfig(**base)


def full_width_params(cfg, seed: int = WEIGHT_SEED, adapter_gain: float = 20.0):
    from oracle.model import init_params
    p = init_params(cfg, seed=seed)
    for k in p:
        if ".adapter." in k:          # larger than the 1e-3 init so that the adapter arithmetic is visible in the outputs
            p[k] = p[k] * adapter_gain
    return p


def full_depth_params(cfg, seed: int = WEIGHT_SEED, adapter_gain: float = 20.0):
    """Weights of the full-DEPTH fixtures (28 blocks = 5.9 B values): as full_width_params but one random stream per
    block, drawn concurrently (oracle.model.init_params(layer_seeds=True): ~20 s instead of ~140 s)."""
    from oracle.model import init_params
    p = init_params(cfg, seed=seed, layer_seeds=True)
    for k in p:
        if ".adapter." in k:
            p[k] = p[k] * adapter_gain
    return p


def oracle_window(caps: torch.Tensor, P: int, eos: int, multiple: int = 64) -> torch.Tensor:
    """The captions the ORACLE is evaluated on in the S = 2048 parity tests: the first T = ceil64(P + longest caption + 2)
    positions.  The product path under test still runs all 2048 positions; for the oracle the rest is dead weight -- the
    loss is masked behind the first eos (reference magma/utils.py:334-364), and under the causal mask a position cannot influence
    an earlier one, so loss, target-row logits and every gradient are the SAME function values on the window (checked
    against the full-length evaluation in tests/test_oracle_pins.py::test_oracle_window_is_exact).  Cuts each fp32 / bf16
    autograd leg of the CPU oracle by ~S / T."""
    longest = 0
    for row in caps:
        nz = (row != eos).nonzero()
        longest = max(longest, int(nz.max()) + 1 if nz.numel() else 0)
    T = min(caps.shape[1], -(-(P + longest + 2) // multiple) * multiple)
    return caps[:, :T].contiguous()


def lm_only(params):
    return {k: v for k, v in params.items() if k.startswith("lm.")}


def greedy_inputs(cfg, seed: int, B: int = GREEDY_B, S0: int = PREFILL_LEN):
    """bf16-representable prefill embeddings (the same values enter the oracle and the HIP path)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, S0, cfg.d_model, generator=g).to(torch.bfloat16).float()


def oracle_greedy_margins(params, cfg, emb, steps: int):
    """Free-running greedy decode of the fp32 oracle -> (tokens (B, S0+steps), per-step min over rows of
    (top1 - top2) / std(logits), per-step logits)."""
    from oracle.model import generate_greedy
    toks, logits = generate_greedy(params, cfg, emb, steps, stop_on_eos=False)
    margins = []
    for lg in logits:
        top2 = torch.topk(lg, 2, dim=-1).values
        margins.append(float(((top2[:, 0] - top2[:, 1]) / lg.std(dim=-1)).min()))
    return toks, margins, logits


GREEDY_INPUT_SEED = 1692
# tests/test_fulldepth_gpu.py: 28 blocks, free-running greedy for FULLDEPTH_STEPS steps (tools/find_margin_seed.py ... 28 8)
FULLDEPTH_STEPS = 8
FULLDEPTH_INPUT_SEED = 3
