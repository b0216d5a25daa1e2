"""CPU oracle for the MAGMA hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain PyTorch on the CPU (fp32 by default), the
arithmetic the reference executes on the path
  CLIP-RN50x16 trunk -> ImagePrefix -> GPT-J-6B blocks with MAGMA adapters
  -> logits / shifted cross-entropy / greedy generate,
plus the integer label builder.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product (``magma_amd``) never
imports ``oracle`` and fails loudly when its HIP library is missing.

PARITY PIN STATUS (see DESIGN.md "Oracle"):
  * adapters / sampling filters / build_labels: pinned against the reference's
    own modules imported from /root/reference (tests/golden/make_golden.py).
  * GPT-J block arithmetic: the reference's implementation lives in the
    un-vendored fork ``finetuneanon/transformers@gpt-neo-localattention3-rp-b``
    (reference setup.py:4).  It is pinned against the independent statement of
    the same published algorithm that IS installed: HF ``GPTJForCausalLM``
    (tests/test_oracle_vs_hf.py).  The reference holds no golden vectors.
  * CLIP ModifiedResNet trunk: un-vendored ``openai/CLIP`` (reference
    setup.py:5), no stand-in available offline -> "parity unpinned" beyond the
    two reference constants (3072 channels, 144 tokens @384^2,
    reference magma/image_prefix.py:13,20) and the 136.2 M parameter count.
"""
