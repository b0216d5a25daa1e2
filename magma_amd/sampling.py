"""Autoregressive sampling loop -- behaviour of reference magma/sampling.py:43-121
(prefill on embeddings, then one token per step with the KV cache handed back;
greedy when temperature == 0.0; temperature / top-k / the reference's own
top-p filter otherwise; early stop when every row emitted EOS).

Differences: token selection -- greedy argmax AND the sampled branch --
runs as HIP kernels inside the captured token step (csrc/sampling.hip); the
per-step ``(next_token == eos).all()`` host sync of the reference (:109) became
a device-side record read every few steps.  ``top_k_filter`` / ``top_p_filter``
below are the host statements of the same rules (pinned to the reference's own
functions, tests/test_oracle_pins.py) and serve LM objects other than the engine."""
import os
from typing import List, Union

import torch
import torch.nn.functional as F


def top_p_filter(logits, threshold: float = 0.9):
    """The reference's own "nucleus" rule, kept literally: sort descending, mark the
    ranks whose cumulative probability is still below ``1 - threshold``, shift that mark one rank to
    the right, never drop rank 0 -- which is NOT textbook top-p (it is usually a no-op)."""
    ranked, order = torch.sort(logits, descending=True)
    below = torch.cumsum(F.softmax(ranked, dim=-1), dim=-1) < (1 - threshold)
    drop = torch.zeros_like(below)
    drop[..., 1:] = below[..., :-1]
    ranked = ranked.masked_fill(drop, float("-inf"))
    return ranked.scatter(1, order, ranked)


def top_k_filter(logits, k):
    """Keep the k largest logits of every row, -inf elsewhere."""
    assert k > 0
    kept_val, kept_idx = torch.topk(logits, k)
    return torch.full_like(logits, float("-inf")).scatter_(1, kept_idx, kept_val)


def remove_tokens_after_eos(tensor, eos_token, image_token):
    """Everything from the first EOS on becomes EOS; image placeholders and EOS are then dropped."""
    hits = (tensor == eos_token).nonzero()
    if hits.any():
        tensor[hits[0]:] = eos_token
    return [tok for tok in tensor.tolist() if tok not in (image_token, eos_token)]


@torch.no_grad()
def generate(model, embeddings, max_steps: int = 100, temperature: float = 0.7, top_k: int = 0,
             top_p: float = 0.9, eos_token: int = None, decode: bool = True,
             stop_on_eos: bool = True, seed: int = None, eos_check_every: int = None) -> Union[List[str], torch.Tensor]:
    """reference sampling.py:43-121.  Token selection (argmax, or top-k / the reference's top-p rule / softmax /
    multinomial) and the ``(next_token == eos).all()`` test run on the device inside the captured token step; the host
    reads the recorded "first all-eos step" every ``eos_check_every`` steps (default 8, MAGMA_EOS_CHECK_EVERY) instead of
    synchronising on every token, and cuts the output there -- same result as the reference's per-step break.
    ``seed`` fixes the sampling stream (default: drawn from torch's CPU generator, so torch.manual_seed reproduces a run)."""
    eos_token = eos_token or model.eos_token
    was_training = model.training
    model.eval()
    b, s, _ = embeddings.shape
    dev = embeddings.device
    out = torch.full((b, s + max_steps), eos_token, dtype=torch.long, device=dev)
    out[:, :s] = model.image_token
    n = s
    past = None
    greedy = temperature == 0.0
    mode = None if greedy else (float(temperature), int(top_k), float(top_p))
    if seed is None and not greedy:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    every = eos_check_every or int(os.environ.get("MAGMA_EOS_CHECK_EVERY", "8"))
    # the HIP engine selects the token itself; any other LM object gets the reference's call (sampling.py:81-93)
    on_device = getattr(model.lm, "device_token_selection", False)
    first_kw = dict(sampling=mode, eos_token=eos_token, seed=seed) if on_device else {}
    step_kw = dict(sampling=mode) if on_device else {}
    for i in range(max_steps):
        if i == 0:
            outputs = model.lm(inputs_embeds=embeddings, use_cache=True, past_key_values=None, cache_hint=max_steps,
                               reuse_cache=True, **first_kw)
        elif on_device:      # the token selected by the previous step is fed back on the device: nothing crosses the host
            outputs = model.lm(input_ids=None, use_cache=True, past_key_values=past, feed_back=True, **step_kw)
        else:
            outputs = model.lm(input_ids=out[:, n - 1:n], use_cache=True, past_key_values=past, **step_kw)
        past = outputs.past_key_values
        if on_device:
            n += 1
            if stop_on_eos and eos_token is not None and ((i + 1) % every == 0 or i + 1 == max_steps):
                first = int(outputs.eos_state[1])            # one host sync per `every` steps
                if first >= 0:
                    n = s + first + 1                        # tokens after the first all-eos step are dropped again
                    break
            continue
        logits = outputs.logits[:, -1, :].float()            # any other LM object: the reference's host-side arithmetic
        if greedy:
            next_token = outputs.next_token.unsqueeze(1) if outputs.get("next_token") is not None else \
                torch.argmax(logits, dim=-1, keepdim=True)
        else:
            if top_k > 0:
                logits = top_k_filter(logits, k=top_k)
            if top_p > 0:
                logits = top_p_filter(logits, threshold=top_p)
            probs = F.softmax(logits / temperature, dim=-1)
            next_token = torch.multinomial(probs, num_samples=1)
        out[:, n:n + 1] = next_token
        n += 1
        if stop_on_eos and eos_token is not None and bool((next_token == eos_token).all()):
            break
    if on_device:            # one copy of the token history the bookkeeping kernel kept
        out[:, s:n] = past.history[:, : n - s]
        model.lm.engine.check_decode(past)
    out = out[:, :n]
    if decode:
        out = [model.tokenizer.decode(remove_tokens_after_eos(row, eos_token, model.image_token)) for row in out]
    model.train(was_training)
    return out
