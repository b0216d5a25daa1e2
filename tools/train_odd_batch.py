"""Training step at batch sizes the benchmark does not use (1, 3, 5) on a full-width 2-block model: finite loss,
finite gradients, loss decreases over 3 steps on a fixed batch."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import Magma
from magma_amd.language_model import GPTJConfig
from magma_amd.datasets import synthetic_batch
from magma_amd.train_engine import MagmaEngine

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Magma("MAGMA_v1", device=dev, lm_config=GPTJConfig(num_layers=2, vocab_size=50258))
model.config.gradient_accumulation_steps = 1
eng = MagmaEngine(model); eng.train()
ok = True
for B in (1, 3, 5):
    images, caps = synthetic_batch(B, 224, model.seq_len, model.eos_token, 50256, 7 + B, device=dev, dtype=torch.bfloat16)
    losses = []
    for _ in range(3):
        o = eng(images, caps); eng.backward(o.loss)
        gn = sum(float(g.grad.float().pow(2).sum()) for g in eng.groups) ** 0.5
        eng.step(); losses.append(float(o.loss))
    good = all(map(lambda v: v == v and abs(v) < 1e4, losses)) and gn == gn and gn > 0
    ok &= good
    print(json.dumps({"B": B, "losses": losses, "grad_norm": gn, "ok": good}), flush=True)
print("ALL_OK" if ok else "FAILED")
