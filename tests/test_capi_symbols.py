"""The C-ABI library loads without a GPU and exports every symbol that
include/*.h declares (no compute call is made here)."""
import ctypes as C
import os
import re

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
