"""The C-ABI library loads without a GPU and exports every symbol that
include/*.h declares (no compute call is made here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared(header):
    with open(os.path.join(ROOT, "include", header)) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+|rtb_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(product_lib):
    names = []
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if h.endswith(".h"):
            names += _declared(h)
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(product_lib, n)]
    assert not missing, missing


def test_python_binding_covers_the_header(product_lib):
    from raytracer_amd import _capi
    declared = set(_declared("rt_amd.h"))
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)


def test_struct_sizes_match_between_binding_and_oracle(product_lib, oracle):
    from raytracer_amd import _capi
    assert C.sizeof(_capi.Pass) == C.sizeof(oracle.Pass)
    assert C.sizeof(_capi.Light) == C.sizeof(oracle.Light)
    assert C.sizeof(_capi.Pass) % 8 == 0


def test_version_and_error_text(product_lib):
    from raytracer_amd import _capi, api
    assert b"gfx950" in product_lib.rt_version()
    try:
        api.Bvh.from_json(api.Mesh([[0, 0, 0, 1, 0, 0, 0, 1, 0]]), "not json")
    except _capi.RtError as e:
        assert e.status == -4 and "JSON" in e.message
    else:
        raise AssertionError("expected RtError")


def _build_c_caller(tmp_path):
    import subprocess
    lib_dir = os.path.join(ROOT, "raytracer_amd", "lib")
    exe = str(tmp_path / "caller")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror",
                           "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "caller.c"),
                           "-L", lib_dir, "-lraytracer_amd", "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


@pytest.mark.gpu
def test_plain_c_caller_on_the_gpu(product_lib, gpu_device, tmp_path):
    """The same C program with its device part: rt_scene_create and
    rt_trace_rays from C on hand-computed rays (exact t, u, v; back-face cull;
    tmax cut), in all three modes."""
    import subprocess
    out = subprocess.run([_build_c_caller(tmp_path), "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)


def test_plain_c_caller(product_lib, tmp_path):
    """tests/c_abi/caller.c: a C99 program (-pedantic -Werror) includes both
    headers, links the library and drives the host-side entry points; the
    struct sizes it prints equal the ctypes mirror's."""
    import subprocess
    from raytracer_amd import _capi
    exe = _build_c_caller(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    words = out.stdout.split()
    sizes = dict(zip(words[0::2], map(int, words[1::2])))
    from raytracer_amd import api
    mirror = {"rt_pass": _capi.Pass, "rt_camera": _capi.Camera, "rt_settings": _capi.Settings,
              "rt_light": _capi.Light, "rt_stats": _capi.Stats, "rt_mesh": _capi.Mesh}
    for name, cls in mirror.items():
        assert sizes[name] == C.sizeof(cls), name
    assert sizes["rt_ray"] == 32 == api.RAY_DTYPE.itemsize and sizes["rt_hit"] == 16 == api.HIT_DTYPE.itemsize
