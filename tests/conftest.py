import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the float64 oracle's op sizes (a few thousand rows) do not use a 256-core host: beyond ~32 threads the intra-op pool
    # only adds fork / join time (the 256-molecule parity case: 186 s with the default pool, 94 s with 32)
    import torch

    if torch.get_num_threads() > 32:
        torch.set_num_threads(32)


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
