"""Device-side token selection (csrc/sampling.hip): the sampled branch of reference
magma/sampling.py:99-109 inside the captured token step.

  * the filters, bit for bit against the outputs of the REFERENCE's own top_k_filter / top_p_filter captured in
    tests/golden/reference_pins.pt, and against the host statements (themselves pinned) at the full vocabulary on peaked,
    flat, tied and already-filtered rows;
  * the multinomial draw against a restatement in Python integers / float64 of the same Philox4x32-10 stream, and its
    empirical frequencies against the probabilities;
  * the generate() loop: captured graph == eager launches for a fixed seed, torch.manual_seed reproducibility, the
    device-side all-eos record against the reference's per-step break."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PINS = torch.load(os.path.join(os.path.dirname(__file__), "golden", "reference_pins.pt"), weights_only=False)


def _filter(logits, top_k, top_p):
    from magma_amd import ops
    x = logits.cuda().float().contiguous()
    f = torch.empty_like(x)
    ops.sample(x, 1.0, top_k, top_p, None, None, filtered=f, want_token=False)
    return f.cpu()


def test_filters_equal_reference_pins(dev):
    lg = PINS["top_p"]["logits"]
    assert torch.equal(_filter(lg, 0, 0.9), PINS["top_p"]["out_0.9"])
    assert torch.equal(_filter(lg, 0, 0.5), PINS["top_p"]["out_0.5"])
    assert torch.equal(_filter(PINS["top_k"]["logits"], 5, 0.0), PINS["top_k"]["out_5"])
    # the two filters chained, as generate() applies them (reference sampling.py:100-103)
    from magma_amd import sampling as S
    want = S.top_p_filter(S.top_k_filter(lg.clone(), 7), 0.9)
    assert torch.equal(_filter(lg, 7, 0.9), want)


@pytest.mark.parametrize("kind", ["peaked", "flat", "random", "bf16_ties"])
def test_filters_full_vocabulary(dev, kind):
    from magma_amd import sampling as S
    V, B = 50258, 6
    g = torch.Generator().manual_seed({"peaked": 1, "flat": 2, "random": 3, "bf16_ties": 4}[kind])
    x = torch.randn(B, V, generator=g)
    if kind == "peaked":
        x = x * 4.0
        x[:, 17] += 30.0                      # top-1 probability > 0.1: the reference's rule is a no-op
    elif kind == "flat":
        x = x * 0.01                          # every probability ~ 2e-5: thousands of ranks are dropped
    elif kind == "bf16_ties":
        x = (x * 0.5).to(torch.bfloat16).float()   # coarse values: many exact ties, also at the boundaries
    for top_k, top_p in ((0, 0.9), (0, 0.3), (40, 0.0), (1000, 0.9)):
        got = _filter(x, top_k, top_p)
        want = x.clone()
        if top_k:
            want = S.top_k_filter(want, top_k)
        if top_p:
            want = S.top_p_filter(want, top_p)
        assert torch.equal(torch.isneginf(got) | (got == x), torch.ones_like(got, dtype=torch.bool))   # only -inf or the original value
        diff = (torch.isneginf(got) != torch.isneginf(want)).sum(1)
        if kind == "bf16_ties":
            # torch.topk / torch.sort keep an unspecified subset of exact ties (at the top-k boundary, at the top-p boundary and
            # among several maxima: which one is "rank 0"); the kernel resolves each by index order.  The SETS may therefore
            # differ among EQUAL VALUES only: per distinct value the number of survivors must agree (+-1 at the top-p boundary
            # for the summation order, as below).
            for b in range(B):
                d = torch.isneginf(got[b]) != torch.isneginf(want[b])
                off = 0
                for v in x[b][d].unique():
                    sel = x[b] == v
                    off += abs(int((~torch.isneginf(got[b][sel])).sum()) - int((~torch.isneginf(want[b][sel])).sum()))
                assert off <= 1, (kind, top_k, top_p, b, off)
        else:
            # summation order (fixed point here, fp32 cumsum there) can move the boundary by one rank at most
            assert int(diff.max()) <= 1, (kind, top_k, top_p, diff.tolist())
        if kind == "flat" and top_p == 0.9 and not top_k:
            assert int(torch.isneginf(got).sum(1).min()) > 1000          # the rule really fired
        if kind == "peaked" and not top_k:
            assert int(torch.isneginf(got).sum()) == 0


def philox4x32_10(c, k0, k1):
    c = [int(v) & 0xFFFFFFFF for v in c]
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c


def restated_draw(row, kept, temperature, seed, step, b):
    """float64 statement of the draw: u = 64 Philox bits / 2^64, first index whose inclusive CDF exceeds u * total."""
    c = philox4x32_10([step, b, 0, 0], seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = ((c[0] << 32) | c[1]) / 2.0 ** 64
    x = row.double().numpy()
    w = np.where(kept.numpy(), np.exp((x - x.max()) / temperature), 0.0)
    cdf = np.cumsum(w)
    t = u * cdf[-1]
    i = int(np.searchsorted(cdf, t, side="right"))
    margin = min(abs(t - cdf[i - 1]) if i > 0 else 1.0, abs(cdf[i] - t)) / cdf[-1]
    return i, margin


def test_multinomial_matches_restatement(dev):
    from magma_amd import ops
    V, B = 50258, 8
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(B, V, generator=g) * 3.0)
    xs = x.cuda()
    seed = 0x1234_5678_9ABC_DEF1
    seed_t = torch.tensor([seed], dtype=torch.int64, device="cuda")
    for (T, k, p) in ((0.7, 0, 0.9), (1.3, 50, 0.0), (1.0, 0, 0.0)):
        f = torch.empty_like(xs)
        for step in (0, 1, 5, 1000):
            state = torch.tensor([step, -1], dtype=torch.int32, device="cuda")
            tok = ops.sample(xs, T, k, p, seed_t, state, filtered=f).cpu()
            kept = ~torch.isneginf(f.cpu())
            for b in range(B):
                want, margin = restated_draw(x[b], kept[b], T, seed, step, b)
                assert bool(kept[b, tok[b]])
                assert int(tok[b]) == want or margin < 1e-6, (T, k, p, step, b, int(tok[b]), want, margin)


def test_multinomial_frequencies(dev):
    """4000 draws (4000 rows) from one 6-way distribution: frequencies within 4.5 sigma of softmax(logits / T)."""
    from magma_amd import ops
    V, N, T = 1000, 4000, 0.8
    row = torch.full((V,), -30.0)
    row[[3, 99, 500, 501, 998, 0]] = torch.tensor([2.0, 1.0, 0.5, 0.0, -0.5, -1.0])
    x = row[None, :].repeat(N, 1).cuda()
    state = torch.tensor([7, -1], dtype=torch.int32, device="cuda")
    seed_t = torch.tensor([42], dtype=torch.int64, device="cuda")
    tok = ops.sample(x, T, 0, 0.0, seed_t, state).cpu()
    probs = torch.softmax(row.double() / T, 0)
    counts = torch.bincount(tok, minlength=V).double()
    for i in (3, 99, 500, 501, 998, 0):
        sd = (N * probs[i] * (1 - probs[i])).sqrt()
        assert abs(counts[i] - N * probs[i]) < 4.5 * sd + 1, (i, float(counts[i]), float(N * probs[i]))
    assert counts.sum() == N and counts[[3, 99, 500, 501, 998, 0]].sum() >= N - 1
    # a different step draws a different stream; the same (seed, step) the same one
    state2 = torch.tensor([8, -1], dtype=torch.int32, device="cuda")
    assert not torch.equal(ops.sample(x, T, 0, 0.0, seed_t, state2).cpu(), tok)
    assert torch.equal(ops.sample(x, T, 0, 0.0, seed_t, state).cpu(), tok)


def test_generate_sampling_graph_equals_eager_and_is_reproducible(dev):
    from magma_amd.testing import build_reduced_magma
    torch.manual_seed(3)
    model = build_reduced_magma(dev)
    model.eval()
    emb = model.embed([torch.randn(2, 3, 64, 64), torch.randint(0, 1000, (2, 5))])
    a = model.generate(emb, max_steps=10, temperature=0.9, top_k=20, top_p=0.9, decode=False, stop_on_eos=False, seed=123)
    b = model.generate(emb, max_steps=10, temperature=0.9, top_k=20, top_p=0.9, decode=False, stop_on_eos=False, seed=123)
    c = model.generate(emb, max_steps=10, temperature=0.9, top_k=20, top_p=0.9, decode=False, stop_on_eos=False, seed=124)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert a.shape == (2, emb.shape[1] + 10)
    # the same call without the captured graph (eager launches of the same token step)
    eng = model.lm.engine
    mode = (0.9, 20, 0.9)
    out = model.lm(inputs_embeds=emb, use_cache=True, cache_hint=10, sampling=mode, eos_token=model.eos_token, seed=123)
    toks, cache = [out.next_token.clone()], out.past_key_values
    for _ in range(9):
        _, tk = eng.decode(toks[-1][:, None], cache, use_graph=False, sampling=mode)
        toks.append(tk.clone())
    assert torch.equal(a[:, emb.shape[1]:], torch.stack(toks, 1))
    # torch.manual_seed reproduces a run when no seed is passed (reference: torch.multinomial follows the global generator)
    torch.manual_seed(77); d1 = model.generate(emb, max_steps=6, temperature=0.7, decode=False, stop_on_eos=False)
    torch.manual_seed(77); d2 = model.generate(emb, max_steps=6, temperature=0.7, decode=False, stop_on_eos=False)
    assert torch.equal(d1, d2)
    strs = model.generate(emb, max_steps=4, temperature=0.7, top_k=5, top_p=0.9)
    assert isinstance(strs, list) and len(strs) == 2


def test_all_eos_record_and_early_stop(dev):
    from magma_amd import ops
    from magma_amd.testing import build_reduced_magma
    state = torch.tensor([0, -1], dtype=torch.int32, device="cuda")
    eos = 9
    seq = [[1, 9], [9, 9], [9, 2], [9, 9]]            # all-eos first at step 1
    for t in seq:
        ops.sample_finish(torch.tensor(t, dtype=torch.int64, device="cuda"), eos, state)
    assert state.tolist() == [4, 1]
    # generate(): the reference breaks right after the step at which every row produced eos (sampling.py:109-110)
    torch.manual_seed(5)
    model = build_reduced_magma(dev)
    model.eval()
    with torch.no_grad():
        model.lm.lm_head.bias[model.eos_token] += 1e4       # every step emits eos
    model.lm.invalidate_packed()
    emb = model.embed([torch.randint(0, 1000, (3, 6))])
    for every in (1, 4, 8):
        toks = model.generate(emb, max_steps=12, temperature=0.0, decode=False, eos_check_every=every)
        assert toks.shape == (3, 6 + 1) and bool((toks[:, -1] == model.eos_token).all())
    toks = model.generate(emb, max_steps=12, temperature=0.8, top_p=0.9, decode=False, seed=1)
    assert toks.shape == (3, 6 + 1)
    assert model.generate(emb, max_steps=5, temperature=0.0, decode=False, stop_on_eos=False).shape == (3, 11)
