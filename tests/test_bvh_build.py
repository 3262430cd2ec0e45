"""Host-side BVH: the O(n log n) builder must reproduce the oracle's literal
restatement of bvh.go:31-212 exactly (topology, bounds, planes, triangle
order); JSON and binary hand-over must round-trip."""
import json
import os

import numpy as np
import pytest

from raytracer_amd import api, scenes
from util import cornell, oracle_scene, product_mesh

KEYS = ("is_leaf", "start", "end", "depth", "bmin", "bmax", "split", "order")


def _same(a, b):
    return all(a[k].shape == b[k].shape and np.array_equal(a[k], b[k]) for k in KEYS)


@pytest.mark.parametrize("n,leaf,depth,seed", [
    (1, 6, 16, 1), (2, 1, 8, 2), (3, 1, 8, 3), (7, 6, 16, 4), (34, 6, 16, 5), (500, 2, 40, 6),
    (4096, 6, 16, 7), (4096, 6, 5, 8), (30000, 8, 40, 9), (2000, 0, 12, 10),
    (140000, 8, 40, 11),      # above the size from which subtrees, axis sweeps and partitions run on threads
])
def test_builder_matches_literal_restatement(product_lib, oracle, n, leaf, depth, seed):
    sc = scenes.soup(n, seed=1000 + seed)
    b = api.Bvh.build(product_mesh(sc), True, leaf, depth)
    o = oracle_scene(oracle, sc, build=(True, leaf, depth))
    assert _same(b.export(), o.export_bvh())


def test_builder_with_duplicate_keys_and_group_indices(product_lib, oracle):
    """Centres that tie along an axis are ordered by Triangle.Index, which
    restarts per OBJ group (context.go:118); full ties fall back to input order
    on both sides."""
    sc = scenes.soup(600, seed=77)
    v = sc["verts"].reshape(-1, 3, 3).copy()
    v[:, :, 0] = np.round(v[:, :, 0] * 4) / 4     # many equal x centres
    v[100:200] = v[0:100]                          # exact duplicates
    sc["verts"] = v.reshape(-1, 9)
    sc["group_index"] = (np.arange(600) % 150).astype(np.int32)
    b = api.Bvh.build(product_mesh(sc), True, 4, 20)
    o = oracle_scene(oracle, sc, build=(True, 4, 20))
    assert _same(b.export(), o.export_bvh())


def test_use_bvh_false_gives_single_leaf_in_input_order(product_lib):
    sc = scenes.soup(50, seed=3)
    e = api.Bvh.build(product_mesh(sc), False, 6, 16).export()
    assert e["is_leaf"].tolist() == [1] and e["end"][0] == 49
    assert e["order"].tolist() == list(range(50))


def test_cornell_tree(product_lib, oracle):
    sc = cornell()
    assert sc["verts"].shape[0] == 34
    b = api.Bvh.build(product_mesh(sc), True, 6, 16)
    o = oracle_scene(oracle, sc, build=(True, 6, 16))
    assert _same(b.export(), o.export_bvh())
    e = b.export()
    # leaves partition the array; interior ranges are the union of their children
    leaves = [(s, t) for l, s, t in zip(e["is_leaf"], e["start"], e["end"]) if l]
    assert sorted(leaves)[0][0] == 0 and sorted(leaves)[-1][1] == 33
    assert sum(t - s + 1 for s, t in leaves) == 34


def test_json_round_trip_and_reference_schema(product_lib):
    sc = scenes.soup(300, seed=11)
    mesh = product_mesh(sc)
    b = api.Bvh.build(mesh, True, 6, 16)
    text = b.to_json()
    doc = json.loads(text)
    root = doc["Root"]
    # field names of models.BVHNode / models.AABB (bvh.go:17-29, aabb.go:11-22)
    assert set(root) == {"Depth", "Index", "LeftChild", "RightChild", "Bounds", "StartIndex",
                         "EndIndex", "SplitPlane"}
    assert set(root["Bounds"]) == {"Position", "Width", "Height", "Length", "Min", "Max", "Bounds"}
    assert root["Index"] == 0 and root["StartIndex"] == 0 and root["EndIndex"] == 299
    b2 = api.Bvh.from_json(mesh, text)
    assert _same(b.export(), b2.export())
    # Go matches field names case-insensitively
    b3 = api.Bvh.from_json(mesh, text.replace('"LeftChild"', '"leftchild"'))
    assert _same(b.export(), b3.export())


def test_json_reader_takes_any_member_order_and_unknown_members(product_lib):
    """The hand-over text is read token by token; what encoding/json accepts must
    still load: members in any order, other spellings of the case, unknown
    members of any shape, Min/Max without the [min,max] pair, white space."""
    sc = scenes.soup(500, seed=12)
    mesh = product_mesh(sc)
    b = api.Bvh.build(mesh, True, 4, 20)
    doc = json.loads(b.to_json())

    def rewrite(n, depth=0):
        if n is None:
            return None
        bounds = dict(n["Bounds"])
        if depth % 2:
            del bounds["Bounds"]                      # Min/Max only
        else:
            bounds = {"extra": {"a": [1, {"b": None}], "c": "}"}, "BOUNDS": bounds["Bounds"]}
        out = {"splitplane": n["SplitPlane"], "EndIndex": n["EndIndex"], "Bounds": bounds,
               "Note": "x\\\"y", "RightChild": rewrite(n["RightChild"], depth + 1),
               "startindex": n["StartIndex"], "LeftChild": rewrite(n["LeftChild"], depth + 1)}
        return out

    text = json.dumps({"Version": [1, 2, {"k": "v"}], "Root": rewrite(doc["Root"])}, indent=1)
    assert _same(b.export(), api.Bvh.from_json(mesh, text).export())


def test_json_errors_are_reported_not_raised_as_crashes(product_lib):
    mesh = product_mesh(scenes.soup(10, seed=1))
    with pytest.raises(api._capi.RtError) as e:
        api.Bvh.from_json(mesh, "{\"Root\": ")
    assert e.value.status == -4
    with pytest.raises(api._capi.RtError):
        api.Bvh.from_json(mesh, "{}")
    leaf = {"Bounds": {"Min": [0, 0, 0], "Max": [1, 1, 1]}, "StartIndex": 0, "EndIndex": 9,
            "LeftChild": None, "RightChild": None, "SplitPlane": [0, 0, 0, 0]}
    assert api.Bvh.from_json(mesh, json.dumps({"Root": leaf})).export()["is_leaf"].tolist() == [1]
    for bad in ({"Root": None}, {"Root": 3}, [leaf],
                {"Root": dict(leaf, LeftChild=leaf)},                       # a single child
                {"Root": dict(leaf, Bounds={"Min": [0, 0, 0]})},            # no usable bounds
                {"Root": dict(leaf, Bounds={"Min": [0, 0, "x"], "Max": [1, 1, 1]})},
                {"Root": dict(leaf, EndIndex=10)},                          # past the triangle array
                {"Root": dict(leaf, EndIndex=1e12)}):
        with pytest.raises(api._capi.RtError):
            api.Bvh.from_json(mesh, json.dumps(bad))
    for text in (json.dumps({"Root": leaf}) + " x", '{"Root": {"Bounds": }}', '{"Root": {"Bounds" 1}}',
                 '{"Root": ' + '{"LeftChild": ' * 5000 + 'null' + '}' * 5000 + '}'):
        with pytest.raises(api._capi.RtError) as e:
            api.Bvh.from_json(mesh, text)
        assert e.value.status == -4


def test_binary_cache_round_trip(product_lib, tmp_path):
    sc = scenes.soup(1000, seed=21)
    mesh = product_mesh(sc)
    b = api.Bvh.build(mesh, True, 6, 24)
    path = str(tmp_path / "soup.bvh")
    b.save(path)
    assert _same(b.export(), api.Bvh.load(mesh, path).export())
    other = product_mesh(scenes.soup(999, seed=21))
    with pytest.raises(api._capi.RtError):
        api.Bvh.load(other, path)
    with open(path, "r+b") as f:
        f.write(b"garbage!")
    with pytest.raises(api._capi.RtError):
        api.Bvh.load(mesh, path)


def test_binary_cache_rejects_stale_or_damaged_files(product_lib, tmp_path):
    """The loader trusts nothing in the file: counts are checked against the
    file size before anything is sized by them, depth is recomputed from the
    preorder leaf flags, ranges and the triangle order are validated."""
    import struct
    sc = scenes.soup(500, seed=22)
    mesh = product_mesh(sc)
    b = api.Bvh.build(mesh, True, 4, 24)
    path = str(tmp_path / "soup.bvh")
    b.save(path)
    good = open(path, "rb").read()
    nn, nt = b.n_nodes, b.n_tris
    hdr = 24
    off = {"is_leaf": hdr, "start": hdr + 4 * nn, "end": hdr + 8 * nn, "depth": hdr + 12 * nn,
           "order": hdr + nn * (16 + 12 + 12 + 16)}

    def load_patched(data):
        q = str(tmp_path / "patched.bvh")
        with open(q, "wb") as f:
            f.write(data)
        return api.Bvh.load(mesh, q)

    def patch(offset, fmt, *values):
        raw = bytearray(good)
        struct.pack_into(fmt, raw, offset, *values)
        return bytes(raw)

    # a stored depth array is not trusted: garbage there changes nothing
    fixed = load_patched(patch(off["depth"], "<3i", 1 << 30, -7, 99)).export()
    assert _same(b.export(), fixed) and fixed["depth"].max() == b.max_depth
    bad = {
        "huge node count": patch(12, "<i", 1 << 30),
        "truncated": good[:-40],
        "trailing bytes": good + b"\0" * 8,
        "leaf flag flipped (tree no longer complete)": patch(off["is_leaf"], "<i", 1),
        "range outside the array": patch(off["end"] + 4 * (nn - 1), "<i", nt + 5),
        "order not a permutation": patch(off["order"], "<2i", 3, 3),
        "negative start": patch(off["start"] + 4, "<i", -1),
    }
    for what, data in bad.items():
        with pytest.raises(api._capi.RtError) as e:
            load_patched(data)
        assert e.value.status == -4, what


def test_mesh_validation(product_lib):
    v = scenes.soup(4, seed=1)["verts"].copy()
    v[2, 4] = np.nan
    with pytest.raises(api._capi.RtError) as e:
        api.Bvh.build(api.Mesh(v), True, 6, 16)
    assert "non-finite" in e.value.message
    with pytest.raises(api._capi.RtError):
        api.Bvh.build(api.Mesh(np.zeros((0, 9), np.float32)), True, 6, 16)
    with pytest.raises(api._capi.RtError):
        api.Bvh.build(api.Mesh(scenes.soup(4, seed=1)["verts"], material=[0, 0, 5, 0]), True, 6, 16)


def test_flatten_layout(product_lib):
    sc = scenes.soup(200, seed=5)
    b = api.Bvh.build(product_mesh(sc), True, 6, 16)
    e = b.export()
    nodes, tris, mat = b.flatten()
    a = nodes[:, 0, 3].view(np.int32)
    c = nodes[:, 1, 3].view(np.int32)
    assert np.array_equal(nodes[:, 0, :3], e["bmin"]) and np.array_equal(nodes[:, 1, :3], e["bmax"])
    leaf = e["is_leaf"] == 1
    assert np.array_equal(a[leaf], e["start"][leaf])
    assert np.array_equal(c[leaf], e["end"][leaf] - e["start"][leaf] + 1)
    assert np.all(c[~leaf] == -1)
    # the jump target of an interior node is the first node after its subtree
    n = len(a)
    for i in np.nonzero(~leaf)[0][:50]:
        j = i + 1
        depth = e["depth"][i]
        while j < n and e["depth"][j] > depth:
            j += 1
        assert a[i] == j
    # triangle records follow the sorted order: (N, V0, E0, -Edge2)
    v = sc["verts"][e["order"]].reshape(-1, 3, 3)
    flat = tris.reshape(-1, 12)
    assert np.array_equal(flat[:, 3:6], v[:, 0])
    assert np.array_equal(flat[:, 6:9], v[:, 1] - v[:, 0])
    assert np.array_equal(flat[:, 9:12], (v[:, 0] - v[:, 2]) * np.float32(-1))
