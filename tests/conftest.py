import sys
from pathlib import Path

import pytest
import torch  # noqa: F401  -- before the HIP library, so one HIP runtime serves both

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) GPU")


@pytest.fixture(scope="session")
def weights_blob():
    import srcnn_cpp_amd
    return srcnn_cpp_amd.load_weights()


@pytest.fixture(scope="session")
def gpu_ctx(weights_blob):
    """A context on cuda:0 with the model loaded.  The HIP library is REQUIRED:
    a missing extension or device is an error, never a skip-to-fallback."""
    import srcnn_cpp_amd
    ctx = srcnn_cpp_amd.Context(0)
    ctx.set_weights_blob(weights_blob)
    yield ctx
    ctx.close()


def pytest_terminal_summary(terminalreporter):
    """The checker reports on itself: how often two runs of the same oracle call disagreed in this process (oracle/__init__.py
    `_forward`: majority of three; met twice in ~11,000 planes on the shared 256-thread hosts of the GPU boxes, never at 64 threads)."""
    mod = sys.modules.get("oracle")
    n = getattr(mod, "anomalies", 0) if mod else 0
    if n:
        terminalreporter.write_line(f"ORACLE ANOMALIES: {n} oracle call(s) needed a third run to settle a disagreement between two runs", yellow=True)
