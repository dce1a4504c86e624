"""pytest configuration: `gpu` marker + shared paths.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI
symbol checks, gloo multi-process sharding.  `-m gpu` runs on the MI355X box:
the HIP path through the C ABI against the oracle and the golden fixtures.
"""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "universal-metal-flash-attention_amd"
for p in (str(ROOT), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / the driver)")
    config.addinivalue_line("markers", "no_gpu_init: the test itself must not initialise the GPU in this process (it starts child processes)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _rearm_fp16_pv(request):
    """GPU tests start with the bf16 forward on its default arithmetic (fp16 P V, option pv_fp16): a test that switched the option
    and failed before restoring it must not hand its setting to the next one.  (Nothing in the library is sticky since round 5:
    V's range is handled per slab on the device)"""
    if request.node.get_closest_marker("gpu") is not None and request.node.get_closest_marker("no_gpu_init") is None:
        try:
            import umfa_torch
            umfa_torch.set_option("pv_fp16", os.environ.get("UMFA_PV_FP16", "1"))
        except Exception:  # noqa: BLE001  (no device: the test itself reports that)
            pass
    yield


@pytest.fixture
def umfa_opts():
    """set launcher switches (umfa_set_option) for one test: umfa_opts(force_w64=1, softmax_reference="exact"); the
    previous values come back at teardown.  (The environment only seeds these switches when the library is first used.)"""
    import umfa_torch
    stack = []

    def setter(**kw):
        ctx = umfa_torch.options(**kw)
        ctx.__enter__()
        stack.append(ctx)

    yield setter
    while stack:
        stack.pop().__exit__(None, None, None)
