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


@pytest.fixture(autouse=True)
def _knot_resolution_is_per_test():
    """A guard that trips doubles the knot counts of the PROCESS (``backend/radial_table._refine``); tests that scale weights to trip
    one must not hand the finer tables to the tests after them."""
    from e3_layers_amd.backend import radial_table as rt

    saved = (rt.KNOTS, rt.KNOTS_SLOPE, rt.KNOTS_MAX, rt.REFINEMENTS)
    for _, g in list(rt._GUARDS.values()):      # (read-backs of an earlier test's model that is still waiting for the collector)
        g.pending.clear()
    yield
    rt.KNOTS, rt.KNOTS_SLOPE, rt.KNOTS_MAX, rt.REFINEMENTS = saved
