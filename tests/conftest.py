import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "probe: a measurement-helper test (device clocks, counters), not an oracle comparison; collected LAST on "
                                       "purpose so that under `-x` it can never cut off a parity test; `-m \"gpu and not probe\"` is the parity suite")
    # A clean checkout has no built artefacts (they are git-ignored): build them once, like __graft_entry__.build().  A library that
    # travelled with the checkout is rebuilt when it was made from other sources than the tree's (content hash recorded in
    # amx_version(), see __graft_entry__.is_stale) or when AMX_FORCE_BUILD=1 asks for a from-scratch build.
    import __graft_entry__
    lib = os.path.join(ROOT, "rasr_amd", "librasr_amd.so")
    orcs = [os.path.join(ROOT, "oracle", n) for n in ("liboracle.so", "liboracle_fma.so")]
    forced = os.environ.get("AMX_FORCE_BUILD", "0") not in ("", "0")
    if forced or not (os.path.exists(lib) and all(os.path.exists(o) for o in orcs)) or __graft_entry__.is_stale():
        __graft_entry__.build()
        os.environ["AMX_FORCE_BUILD"] = "0"   # once per session (pytest-xdist workers inherit the environment)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # measurement helpers run after every parity test, whatever file they live in (stable sort: everything else keeps its order)
    items.sort(key=lambda it: 1 if "probe" in it.keywords else 0)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def ctx():
    """amx context on cuda:0; fails loudly (no fallback) if the HIP library is missing"""
    import rasr_amd
    c = rasr_amd.Context(0)
    yield c
    c.close()
