import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than ~30 s")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _release_cached_gpu_memory():
    """After every test: hand the caching allocator's free blocks back to the driver.  The multi-process tests (2-8 workers on
    the ONE test GPU, each with its own model) need that memory; late in the suite this process's cache alone was tens of GB."""
    yield
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.empty_cache()


@pytest.fixture
def lib_options():
    """``lib_options(gemm_splitk=0, ...)``: set options of the HIP library (ops.options, include/bya.h bya_set_option) until the
    end of the test -- what ``monkeypatch.setenv("BYA_...")`` did while the C entry points still read the environment."""
    from bind_your_avatar_implementation_amd import ops
    stack = []

    def set_(**kw):
        cm = ops.options(**kw)
        cm.__enter__()
        stack.append(cm)

    yield set_
    for cm in reversed(stack):
        cm.__exit__(None, None, None)


def rel_fro(a, b):
    """relative Frobenius error of a against reference b (both converted to fp32/fp64 on CPU)."""
    import torch
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
