"""smoke(): one small invocation of the hot path on cuda:0 (image -> prefix ->
prefill -> 4 greedy decode steps, reduced-size MAGMA_v1 structure) checked
against the CPU oracle.  Lives OUTSIDE the product package (next to __graft_entry__.py): it imports the oracle, which is the checker."""
import torch


def run_smoke():
    from oracle.model import OracleConfig, embed, generate_greedy, init_params
    from magma_amd.lib import load
    from magma_amd.testing import build_reduced_magma

    load()
    assert torch.cuda.is_available(), "smoke() needs an MI355X"
    dev = torch.device("cuda:0")
    cfg = OracleConfig.tiny()
    params = init_params(cfg, seed=3)
    model = build_reduced_magma(dev)
    model.load_checkpoint_state(params)
    model.eval()
    g = torch.Generator().manual_seed(0)
    images = torch.randn(2, 3, 64, 64, generator=g)
    ids = torch.randint(0, 1000, (2, 5), generator=g)
    emb = model.embed([images, ids])
    ref_emb = embed(params, cfg, [images, ids])
    err = float((emb.float().cpu() - ref_emb).norm() / ref_emb.norm())
    assert err < 3e-2, f"prefix/embedding mismatch vs oracle: rel-L2 {err}"
    toks = model.generate(emb, max_steps=4, temperature=0.0, decode=False, stop_on_eos=False).cpu()
    ref_toks, ref_logits = generate_greedy(params, cfg, ref_emb, 4, stop_on_eos=False)
    # token ids must agree wherever the oracle's own top-1/top-2 margin exceeds bf16 noise
    for step, lg in enumerate(ref_logits):
        top2 = torch.topk(lg, 2, dim=-1).values
        safe = (top2[:, 0] - top2[:, 1]) > 0.05 * lg.std()
        pos = ref_toks.shape[1] - len(ref_logits) + step
        assert bool((toks[safe, pos] == ref_toks[safe, pos]).all()), f"greedy token mismatch at step {step}"
        if not bool((toks[:, pos] == ref_toks[:, pos]).all()):
            break   # a tie flipped: later steps follow a different prefix
    print(f"smoke OK: prefix rel-L2 {err:.2e}, tokens {toks[:, -4:].tolist()} (oracle {ref_toks[:, -4:].tolist()})")
