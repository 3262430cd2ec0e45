// flatten.cpp -- turns the pointer-shaped BVH of the reference
// (models/bvh.go:13-29: BVHNode with child pointers and a Bounds pointer;
// models/triangle.go:12-31: []*Triangle) into the two flat, 16-byte-aligned
// record arrays the kernels read (layout: rt_core.h, DevScene).
#include <cstring>

#include "rt_core.h"
#include "rt_internal.h"

namespace rt {

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
    return RT_OK;
}

}  // namespace rt
