import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the native library normally arrives prebuilt (python -c "import __graft_entry__ as g; g.build()"); build it if
    # this checkout has not been built yet — hipcc cross-compiles without a GPU
    lib = os.path.join(ROOT, "graphvite_amd", "libgvk.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "graphvite_amd", "csrc")])


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle_lib import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref is not built and /root/reference is not present")
    return Reference()


@pytest.fixture(scope="session")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from graphvite_amd.kernels import HipKernels
    return HipKernels()
