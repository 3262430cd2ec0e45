// flatten.cpp -- turns the pointer-shaped BVH of the reference
// (models/bvh.go:13-29: BVHNode with child pointers and a Bounds pointer;
// models/triangle.go:12-31: []*Triangle) into the two flat, 16-byte-aligned
// record arrays the kernels read (layout: rt_core.h, DevScene).
#include <cstring>
#include <limits>

#include "rt_core.h"
#include "rt_internal.h"

namespace rt {

namespace {

// Child-pair records for the ordered traversal.  The reference's leaves hold
// up to BVHMaxLeafSize triangles; here a leaf with more than kFastSplitLeaf is
// cut in two by array position (the range is sorted along its parent's split
// axis, bvh.go:73-76, so the halves are spatially separate) under synthetic
// interior records with tight boxes, until every leaf is small: triangle tests
// are the poorly-vectorised part of the traversal.  Array positions, and with
// them hit ids and tie-breaking, are unchanged.  An empty leaf (the reference
// can create one, bvh.go:94-100 with BVHMaxLeafSize 1) gets NaN bounds that no
// ray enters.
struct PairBuilder {
    const rt_mesh &mesh;
    const Bvh &bvh;
    const std::vector<int32_t> &skip;
    std::vector<float> &pairs;
    const float nan_;

    uint32_t new_record() {
        pairs.resize(pairs.size() + 16, 0.0f);
        return (uint32_t)(pairs.size() / 16 - 1);
    }
    void set_child(uint32_t rec, int side, const float mn[3], const float mx[3], uint32_t ref) {
        float *p = &pairs[(size_t)rec * 16 + (size_t)side * 6];
        for (int k = 0; k < 3; ++k) {
            p[2 * k] = mn[k];
            p[2 * k + 1] = mx[k];
        }
        p = &pairs[(size_t)rec * 16];
        p[12 + side] = rtc::i2f((int32_t)ref);
    }
    // bounds of the triangles at array positions [start, start+count)
    void range_box(int32_t start, int32_t count, float mn[3], float mx[3]) const {
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::numeric_limits<float>::max();
            mx[k] = -std::numeric_limits<float>::max();
        }
        for (int32_t i = start; i < start + count; ++i) {
            const float *v = mesh.verts + 9 * (size_t)bvh.order[(size_t)i];
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) {
                    const float c = v[3 * j + k];
                    if (c < mn[k]) mn[k] = c;
                    if (c > mx[k]) mx[k] = c;
                }
        }
    }
    // reference to a triangle range
    uint32_t leaf_ref(int32_t start, int32_t count) {
        if (count <= rtc::kFastSplitLeaf)
            return rtc::kLeafBit | ((uint32_t)count << 24) | (uint32_t)start;
        const uint32_t rec = new_record();
        const int32_t half = count / 2;
        float mn[3], mx[3];
        const uint32_t l = leaf_ref(start, half);
        range_box(start, half, mn, mx);
        set_child(rec, 0, mn, mx, l);
        const uint32_t r = leaf_ref(start + half, count - half);
        range_box(start + half, count - half, mn, mx);
        set_child(rec, 1, mn, mx, r);
        return rec;
    }
    uint32_t child_ref(int32_t node, float mn[3], float mx[3]) {
        std::memcpy(mn, &bvh.bmin[3 * (size_t)node], 12);
        std::memcpy(mx, &bvh.bmax[3 * (size_t)node], 12);
        if (!bvh.is_leaf[(size_t)node]) return record_of[(size_t)node];
        const int32_t start = bvh.start[(size_t)node];
        const int32_t count = bvh.end[(size_t)node] - start + 1;
        if (count <= 0) {
            // NaN bounds fail every comparison of the slab test
            for (int k = 0; k < 3; ++k) { mn[k] = nan_; mx[k] = nan_; }
            return rtc::kLeafBit;   // count 0
        }
        return leaf_ref(start, count);
    }
    std::vector<uint32_t> record_of;   // interior node -> record index
};

void build_pairs(const rt_mesh &mesh, const Bvh &bvh, const std::vector<int32_t> &skip, FlatScene &out) {
    out.pairs.clear();
    out.fast_ok = false;
    out.fast_root = 0;
    if (mesh.n_tris > rtc::kFastMaxTris || bvh.max_depth > 60) return;
    PairBuilder b{mesh, bvh, skip, out.pairs, std::numeric_limits<float>::quiet_NaN(), {}};
    const int32_t nn = bvh.n_nodes;
    b.record_of.assign((size_t)nn, 0);
    for (int32_t i = 0; i < nn; ++i)
        if (!bvh.is_leaf[(size_t)i]) b.record_of[(size_t)i] = b.new_record();
    float mn[3], mx[3];
    for (int32_t i = 0; i < nn; ++i) {
        if (bvh.is_leaf[(size_t)i]) continue;
        const int32_t l = i + 1, r = skip[(size_t)l];
        const uint32_t rec = b.record_of[(size_t)i];
        const uint32_t lref = b.child_ref(l, mn, mx);
        b.set_child(rec, 0, mn, mx, lref);
        const uint32_t rref = b.child_ref(r, mn, mx);
        b.set_child(rec, 1, mn, mx, rref);
    }
    out.fast_root = b.child_ref(0, mn, mx);   // the root is entered without a box test
    if (out.fast_root == rtc::kLeafBit) {     // a tree whose root is an empty leaf has nothing to traverse in order
        out.pairs.clear();
        out.fast_root = 0;
        return;
    }
    if (out.pairs.empty()) out.pairs.resize(16, 0.0f);
    out.fast_ok = true;
}

}  // namespace

rt_status flatten(const rt_mesh &mesh, const Bvh &bvh, FlatScene &out) {
    using rtc::f4;
    using rtc::i2f;
    if (bvh.n_tris != mesh.n_tris)
        return fail(RT_ERR_INVALID, "BVH was built for a different triangle count");
    const int32_t nn = bvh.n_nodes;
    if (nn <= 0) return fail(RT_ERR_INVALID, "BVH has no nodes");
    // skip link = index of the first node after the subtree, from the preorder
    // sequence of leaf flags
    std::vector<int32_t> skip((size_t)nn);
    {
        std::vector<int32_t> open, pending;
        for (int32_t i = 0; i < nn; ++i) {
            open.push_back(i);
            pending.push_back(bvh.is_leaf[(size_t)i] ? 0 : 2);
            while (!open.empty() && pending.back() == 0) {
                skip[(size_t)open.back()] = i + 1;
                open.pop_back();
                pending.pop_back();
                if (!pending.empty()) pending.back()--;
            }
        }
        if (!open.empty()) return fail(RT_ERR_INVALID, "malformed preorder BVH");
    }
    out.nodes.resize((size_t)nn * 8);
    f4 *nodes = reinterpret_cast<f4 *>(out.nodes.data());
    for (int32_t i = 0; i < nn; ++i) {
        const float *mn = &bvh.bmin[3 * (size_t)i], *mx = &bvh.bmax[3 * (size_t)i];
        int32_t a, b;
        if (bvh.is_leaf[(size_t)i]) {
            a = bvh.start[(size_t)i];
            b = bvh.end[(size_t)i] - bvh.start[(size_t)i] + 1;
            if (b < 0) b = 0;
            if (b > 0 && (a < 0 || (int64_t)a + b > mesh.n_tris))
                return fail(RT_ERR_INVALID, "BVH leaf range outside the triangle array");
            if (b == 0) a = 0;
        } else {
            a = skip[(size_t)i];
            b = -1;
        }
        nodes[2 * (size_t)i] = f4{mn[0], mn[1], mn[2], i2f(a)};
        nodes[2 * (size_t)i + 1] = f4{mx[0], mx[1], mx[2], i2f(b)};
    }
    out.tris.resize((size_t)mesh.n_tris * 12);
    out.tri_mat.resize((size_t)mesh.n_tris);
    out.tri_uv.clear();
    if (mesh.texcoords) out.tri_uv.resize((size_t)mesh.n_tris * 6);
    f4 *tris = reinterpret_cast<f4 *>(out.tris.data());
    for (int32_t pos = 0; pos < mesh.n_tris; ++pos) {
        const int32_t id = bvh.order[(size_t)pos];
        if (id < 0 || id >= mesh.n_tris) return fail(RT_ERR_INVALID, "BVH order entry out of range");
        TriRecord rec;
        make_tri_record(mesh.verts + 9 * (size_t)id, rec);
        tris[3 * (size_t)pos] = f4{rec.normal[0], rec.normal[1], rec.normal[2], rec.v0[0]};
        tris[3 * (size_t)pos + 1] = f4{rec.v0[1], rec.v0[2], rec.e0[0], rec.e0[1]};
        tris[3 * (size_t)pos + 2] = f4{rec.e0[2], rec.v0v2[0], rec.v0v2[1], rec.v0v2[2]};
        const int32_t mat = mesh.material ? mesh.material[id] : 0;
        out.tri_mat[(size_t)pos] = mat | (mesh.is_light[mat] ? rtc::kLightBit : 0);
        if (mesh.texcoords) std::memcpy(&out.tri_uv[6 * (size_t)pos], mesh.texcoords + 6 * (size_t)id, 24);
    }
    build_pairs(mesh, bvh, skip, out);
    return RT_OK;
}

}  // namespace rt
