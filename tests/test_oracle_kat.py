"""Pins the oracle: every known-answer case the reference's own tests hold for
the hot path (models/aabb_test.go), plus build-generated triangle cases."""
import json
import math
import os
import struct

import numpy as np
import pytest

from util import GOLDEN


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.mark.parametrize("mode", [0, 1])
def test_aabb_reference_kats(oracle, mode):
    kat = _load("aabb_kat.json")
    bmin, bmax = kat["box"]["min"], kat["box"]["max"]
    for case in kat["cases"]:
        o = np.array(case["origin"], np.float32)
        # axis.Mul(-1): float32 negation, gives -0 components (aabb_test.go:64)
        d = o * np.float32(-1) if case.get("dir_is_negated_origin") else np.array(case["dir"], np.float32)
        hit, t0, t1 = oracle.aabb_intersect(bmin, bmax, o, d, mode)
        assert hit == case["hit"], case["name"]
        if "tmin" in case:
            assert t0 == case["tmin"] and t1 == case["tmax"], case["name"]


@pytest.mark.parametrize("mode", [0, 1])
def test_triangle_kats(oracle, mode):
    kat = _load("tri_kat.json")
    tri = kat["triangle"]
    for case in kat["cases"]:
        t, u, v = oracle.tri_intersect(tri["v0"], tri["v1"], tri["v2"], case["origin"],
                                       case["dir"], mode)
        assert (t, u, v) == (case["t"], case["u"], case["v"]), case["name"]


def test_slab_nan_path_hits(oracle):
    """A ray lying exactly in a slab face multiplies 0 by Inf; every comparison
    with the NaN is false, so the box is reported hit (aabb_test.go:24-36)."""
    hit, _, _ = oracle.aabb_intersect([-.5] * 3, [.5] * 3, [1, 0.5, 0], [-1, 0, 0])
    assert hit
    hit, _, _ = oracle.aabb_intersect([-.5] * 3, [.5] * 3, [1, -0.5, 0.5], [-1, 0, 0])
    assert hit


def _ulps(a, b):
    ia = struct.unpack("<q", struct.pack("<d", a))[0]
    ib = struct.unpack("<q", struct.pack("<d", b))[0]
    return abs(ia - ib)


def test_go_math_against_libm(oracle):
    """The restated Go kernels (Cephes sin/cos, FreeBSD exp/log, Go's pow) must
    agree with libm to a few ulps; a mis-stated coefficient shows up as
    thousands."""
    L = oracle.lib()
    rng = np.random.default_rng(7)
    for x in rng.uniform(-4, 4, 4000):
        assert _ulps(L.orc_go_cos(x), math.cos(x)) <= 2
    for x in rng.uniform(-30, 30, 4000):
        assert _ulps(L.orc_go_exp(x), math.exp(x)) <= 2
    for x in rng.uniform(1e-9, 1e3, 4000):
        assert _ulps(L.orc_go_log(x), math.log(x)) <= 2
    for x in rng.uniform(0, 4, 4000):
        assert _ulps(L.orc_go_pow(x, 1 / 2.2), math.pow(x, 1 / 2.2)) <= 4
    assert L.orc_go_pow(2.0, 10.0) == 1024.0
    assert L.orc_go_pow(9.0, 0.5) == 3.0
    assert L.orc_go_pow(0.0, 0.4545) == 0.0
    assert L.orc_go_pow(1.0, 123.0) == 1.0
    assert math.isnan(L.orc_go_pow(-2.0, 0.5)) or L.orc_go_pow(-2.0, 0.5) != L.orc_go_pow(-2.0, 0.5)
    # math.Max / math.Min special cases
    assert math.isnan(L.orc_go_max(float("nan"), 0.0))
    assert math.copysign(1, L.orc_go_max(-0.0, 0.0)) == 1.0
    assert math.copysign(1, L.orc_go_min(-0.0, 0.0)) == -1.0


def test_resolve_truncates_and_clamps(oracle):
    acc = np.array([[0.0, 0.5, 2.0], [1.0, 0.25, float("nan")]], np.float32)
    out = oracle.resolve(acc, 1, False, 2.2)
    assert out.tolist() == [[0, 127, 255, 255], [255, 63, 0, 255]]
    out = oracle.resolve(acc * 4, 4, True, 2.2)
    # 0.5^(1/2.2) = 0.7297..., *255 = 186.08 -> 186
    assert out[0].tolist() == [0, 186, 255, 255]


def test_oracle_traversal_first_visited_wins_ties(oracle):
    """Two coplanar identical triangles: the one earlier in the array wins
    (strict '<' at bvh.go:125)."""
    tri = [0, 0, 0, 1, 0, 0, 0, 1, 0]
    s = oracle.Scene(np.array([tri, tri], np.float32))
    s.build_bvh(True, 6, 16)
    r = s.trace_rays(np.array([[0.25, 0.25, 1, 0, 0, -1]], np.float32))
    assert r["tri"][0] == 0 and r["t"][0] == 1.0


def test_rng_is_uniform_24bit(oracle):
    L = oracle.lib()
    s = L.orc_rng_stream(5, 17)
    xs = np.array([L.orc_rng_f32(s, i) for i in range(20000)])
    assert xs.min() >= 0 and xs.max() < 1
    assert abs(xs.mean() - 0.5) < 0.01
    assert np.all(xs * 2 ** 24 == np.floor(xs * 2 ** 24))
    assert L.orc_rng_stream(5, 18) != s and L.orc_rng_stream(6, 17) != s
