import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), compiled on demand with gcc."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def product_lib():
    """libraytracer_amd.so: rebuilt with hipcc only when its sources changed
    (the GPU box receives the prebuilt library with the tree)."""
    from raytracer_amd import build, _capi
    if build.is_stale():
        build.build_library()
    return _capi.lib()


@pytest.fixture(scope="session")
def emul():
    """CPU driver around the kernels' arithmetic header (tests/host_emul)."""
    import ctypes as C
    src = os.path.join(ROOT, "tests", "host_emul", "emul.cpp")
    core = os.path.join(ROOT, "raytracer_amd", "csrc", "rt_core.h")
    gom = os.path.join(ROOT, "raytracer_amd", "csrc", "gomath.h")
    out_dir = os.path.join(ROOT, "tests", "_build")
    out = os.path.join(out_dir, "libemul.so")
    os.makedirs(out_dir, exist_ok=True)
    if not os.path.exists(out) or any(os.path.getmtime(p) > os.path.getmtime(out)
                                      for p in (src, core, gom)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                               "-fPIC", "-shared", "-o", out, src])
    return C.CDLL(out)


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return 0
