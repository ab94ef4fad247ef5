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
