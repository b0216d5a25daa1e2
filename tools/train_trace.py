"""Two MAGMA_v1 training steps (B=16, S=2048) for rocprofv3 --kernel-trace: one warm-up step, then a marker launch
(advance_pos_kernel) so that tools/trace_summary.py <dir> advance_pos can total only the steady-state steps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import Magma, ops
from magma_amd.datasets import synthetic_batch
from magma_amd.train_engine import MagmaEngine

dev = torch.device("cuda:0")
model = Magma(os.environ.get("CFG", "MAGMA_v1"), device=dev)
model.config.gradient_accumulation_steps = 1
eng = MagmaEngine(model); eng.train()
B, S = int(os.environ.get("TB", 16)), model.seq_len
images, caps = synthetic_batch(B, 224, S, model.eos_token, 50256, 1234, device=dev, dtype=torch.bfloat16)
caps_host = caps.cpu()
def one():
    o = eng(images, caps, captions_host=caps_host); eng.backward(o.loss); eng.step()
one(); torch.cuda.synchronize()
marker = torch.zeros(1, dtype=torch.int32, device=dev)
ops.advance_pos(marker, 1); torch.cuda.synchronize()
N = int(os.environ.get("STEPS", 2))
import time
t0 = time.perf_counter()
for _ in range(N): one()
torch.cuda.synchronize()
print("ms_per_step", (time.perf_counter() - t0) * 1e3 / N)
