"""Autoregressive sampling loop -- behaviour of reference magma/sampling.py:43-121
(prefill on embeddings, then one token per step with the KV cache handed back;
greedy when temperature == 0.0; temperature / top-k / the reference's own
top-p filter otherwise; early stop when every row emitted EOS).

Differences: greedy argmax runs in the HIP kernel inside the decode
graph; the per-step ``(next_token == eos).all()`` host sync of the reference
(:109) is kept but can be disabled with ``stop_on_eos=False`` for timing."""
from typing import List, Union

import torch
import torch.nn.functional as F


def top_p_filter(logits, threshold: float = 0.9):
    """Kept literally as published (descending sort, removes
    cum_probs < 1-threshold shifted by one -- not textbook nucleus)."""
    s_logits, s_idx = torch.sort(logits, descending=True)
    cum = torch.cumsum(F.softmax(s_logits, dim=-1), dim=-1)
    remove = cum < (1 - threshold)
    remove[..., 1:] = remove[..., :-1].clone()
    remove[..., 0] = 0
    s_logits[remove] = float("-inf")
    return s_logits.scatter(1, s_idx, s_logits)


def top_k_filter(logits, k):
    assert k > 0
    val, ind = torch.topk(logits, k)
    out = torch.full_like(logits, float("-inf"))
    out.scatter_(1, ind, val)
    return out


def remove_tokens_after_eos(tensor, eos_token, image_token):
    eos_index = (tensor == eos_token).nonzero()
    if eos_index.any():
        tensor[eos_index[0]:] = eos_token
    return [i for i in tensor.tolist() if i != image_token and i != eos_token]


@torch.no_grad()
def generate(model, embeddings, max_steps: int = 100, temperature: float = 0.7, top_k: int = 0,
             top_p: float = 0.9, eos_token: int = None, decode: bool = True,
             stop_on_eos: bool = True) -> Union[List[str], torch.Tensor]:
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
    for i in range(max_steps):
        if i == 0:
            outputs = model.lm(inputs_embeds=embeddings, use_cache=True, past_key_values=None, cache_hint=max_steps,
                               reuse_cache=True)
        else:
            outputs = model.lm(input_ids=out[:, n - 1:n], use_cache=True, past_key_values=past)
        past = outputs.past_key_values
        if greedy and outputs.get("next_token") is not None:
            next_token = outputs.next_token.unsqueeze(1)          # argmax kernel inside the decode graph
        else:
            logits = outputs.logits[:, -1, :].float()
            if greedy:
                from . import ops
                next_token = ops.argmax(logits.contiguous()).unsqueeze(1)
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
    out = out[:, :n]
    if decode:
        out = [model.tokenizer.decode(remove_tokens_after_eos(row, eos_token, model.image_token)) for row in out]
    model.train(was_training)
    return out
