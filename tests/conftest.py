import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def emu_lib():
    """The product sources compiled against the functional HIP emulation (tests/emu) -- kernel-logic checks only."""
    from gnark_amd import _lib
    alt = os.environ.get("GA_EMU_LIB_PATH")   # another build of the same emulation library (sanitizer builds, tools/emu_sanitize.sh)
    if alt:
        return _lib.Library(alt)
    so = os.path.join(ROOT, "tests", "emu", "libgnark_amd_emu.so")
    r = subprocess.run([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(so):
        pytest.fail("emulation build failed:\n" + r.stdout[-2000:] + r.stderr[-4000:])
    return _lib.Library(so)


@pytest.fixture(scope="session")
def emu_ctx(emu_lib):
    from gnark_amd.device import Context
    ctx = Context(0, lib=emu_lib)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def gpu_ctx():
    """Real device context through the hipcc-built libgnark_amd.so (fails loudly if it is missing)."""
    from gnark_amd.device import Context
    ctx = Context(0)
    yield ctx
    ctx.close()
