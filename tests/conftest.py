import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The fp64 oracle is most of the suite's wall time, and torch's default on the GPU box's host (128 threads on 256 logical CPUs) is its
    # slowest setting: the lmax-3 UNet restatement on 4 096 points takes 19.2 s at the default, 8.2 s with 64 threads, 4.9 s with 32, 4.8 s with 16,
    # 5.2 s with 8 (profiles/r05b_oracle_threads.log).  DEDF_TEST_THREADS overrides.
    try:
        import torch
        torch.set_num_threads(int(os.environ.get("DEDF_TEST_THREADS", min(16, os.cpu_count() or 1))))
    except ImportError:
        pass


@pytest.fixture(scope="session")
def built_lib():
    import __graft_entry__ as g
    g.build()
    from diffusion_edf_amd import _lib
    return _lib.load()
