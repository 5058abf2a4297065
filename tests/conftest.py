import os
import sys

import pytest

try:  # torch first: its HIP runtime and libmcl3dl_hip.so's must be the same instance when device pointers are shared
    import torch  # noqa: F401
except Exception:  # pragma: no cover - CPU-only environments without torch
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")
    # A fresh checkout has no built artefacts (they are git-ignored): build them once so the suite is self-contained.
    needed = [os.path.join(ROOT, "mcl_3dl_amd", "libmcl3dl_hip.so"), os.path.join(ROOT, "oracle", "libmcl3dl_oracle.so")]
    if not all(os.path.exists(p) for p in needed):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle_kind():
    """Prefer the real reference (oracle/_ref) when it has been built; the plain-C port otherwise."""
    from oracle import pyoracle
    if pyoracle.available("ref"):
        return "ref"
    if pyoracle.available("port"):
        return "port"
    pytest.fail("no oracle library built: run `python -c 'import __graft_entry__ as g; g.build()'`")


@pytest.fixture(scope="session")
def engine():
    """The HIP engine on cuda:0. Fails loudly (never skips, never falls back) when the extension or GPU is missing."""
    from mcl_3dl_amd import capi
    eng = capi.Engine(0)
    yield eng
    eng.close()
