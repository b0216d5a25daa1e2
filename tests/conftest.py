import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The CPU oracle runs through PyTorch's CPU kernels.  On the GPU boxes' 256-thread hosts the default thread count makes every
# small op (the reduced-size parity tests are made of them) pay a 256-way fork / join -- the same suite took 610 s on one box and
# 820 s on another, tiny-model tests 5 x apart.  32 threads (one CCD group; what bench.py's cpu_baseline found fastest for the
# full-size GEMMs too) is set before torch starts its pools; subprocesses (bench.py self-launch, loader workers) inherit it.
os.environ.setdefault("OMP_NUM_THREADS", str(min(32, os.cpu_count() or 8)))
os.environ.setdefault("MKL_NUM_THREADS", os.environ["OMP_NUM_THREADS"])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    try:
        import torch
        torch.set_num_threads(int(os.environ["OMP_NUM_THREADS"]))
    except Exception:  # noqa: BLE001
        pass


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
