"""us per launch of the device-side sampler (B = 8 rows of 50 258 logits, reference defaults temperature 0.7 / top_p 0.9)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from magma_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, x in (("random_x3", torch.randn(8, 50258, device=dev) * 3.0), ("peaked", torch.randn(8, 50258, device=dev) * 2.0 + 12.0 * (torch.arange(50258, device=dev) == 17)),
                ("flat", torch.randn(8, 50258, device=dev) * 0.01)):
    seed = torch.tensor([1234], dtype=torch.int64, device=dev)
    state = torch.zeros(2, dtype=torch.int32, device=dev)
    tok = torch.empty(8, dtype=torch.int64, device=dev)
    fn = lambda: ops.sample(x, 0.7, 0, 0.9, seed, state, out=tok)
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(name, "us per launch", round(e0.elapsed_time(e1) / 200 * 1e3, 1))
