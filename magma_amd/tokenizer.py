"""Tokenizer for the MAGMA path.

The reference calls ``GPT2TokenizerFast.from_pretrained("gpt2")`` (network) and
adds ``<|image|>`` as cls token (reference magma/utils.py:43-58): ids eos = pad
= 50256, image = 50257, len = 50258.  With no network we (1) use the real GPT-2
tokenizer when its files are available locally (``MAGMA_TOKENIZER_DIR`` or the
HF cache), else (2) fall back -- with a warning -- to a byte-level tokenizer
with the same special ids and vocabulary size, which is enough for synthetic-data
runs and for every shape/plumbing contract on the hot path (token ids are just
integers to it).  The stand-in produces the WRONG ids for real MAGMA weights, so
``Magma.from_checkpoint`` refuses it unless ``MAGMA_ALLOW_BYTE_TOKENIZER=1``."""
from __future__ import annotations

import os
import warnings
from typing import List

import torch

EOS_ID, IMAGE_ID, VOCAB = 50256, 50257, 50258


class ByteTokenizer:
    """utf-8 bytes -> ids 0..255; specials as GPT-2 + <|image|>."""

    eos_token = "<|endoftext|>"
    cls_token = "<|image|>"
    eos_token_id = EOS_ID
    cls_token_id = IMAGE_ID
    pad_token_id = EOS_ID
    padding_side = "right"

    def __init__(self, sequence_length: int = 2048):
        self.model_max_length = sequence_length

    def __len__(self):
        return VOCAB

    def encode(self, text: str, return_tensors=None, max_length=None, padding=None, truncation=False):
        ids: List[int] = list(text.encode("utf-8"))
        if truncation and max_length:
            ids = ids[:max_length]
        if padding == "max_length" and max_length:
            ids = ids + [self.pad_token_id] * (max_length - len(ids))
        if return_tensors == "pt":
            return torch.tensor([ids], dtype=torch.int64)
        return ids

    def decode(self, ids) -> str:
        if torch.is_tensor(ids):
            ids = ids.tolist()
        return bytes(i for i in ids if 0 <= i < 256).decode("utf-8", errors="replace")


def get_tokenizer(name: str = "gpt2", sequence_length: int = 2048):
    if name != "gpt2":
        raise ValueError(f"Tokenizer {name} not recognized")
    local = os.environ.get("MAGMA_TOKENIZER_DIR")
    try:
        from transformers import GPT2TokenizerFast
        tok = GPT2TokenizerFast.from_pretrained(local or "gpt2", local_files_only=True)
        tok.pad_token_id = tok.eos_token_id
        tok.padding_side = "right"
        tok.model_max_length = sequence_length
        tok.add_special_tokens({"cls_token": "<|image|>"})
        if tok.eos_token_id != EOS_ID or tok.cls_token_id != IMAGE_ID or len(tok) != VOCAB:
            raise RuntimeError("local gpt2 tokenizer files are missing or incomplete")
        return tok
    except Exception as e:  # noqa: BLE001 -- no local GPT-2 files: byte-level stand-in, loudly
        if os.environ.get("MAGMA_REQUIRE_GPT2_TOKENIZER") == "1":
            raise RuntimeError("the GPT-2 tokenizer files are not available locally (set MAGMA_TOKENIZER_DIR)") from e
        warnings.warn("GPT-2 tokenizer files not found locally (%s: %s): using the byte-level stand-in (ids 0..255 + "
                      "specials).  Fine for synthetic data; WRONG for real MAGMA checkpoints -- set MAGMA_TOKENIZER_DIR."
                      % (type(e).__name__, str(e)[:120]), RuntimeWarning, stacklevel=2)
        return ByteTokenizer(sequence_length)
