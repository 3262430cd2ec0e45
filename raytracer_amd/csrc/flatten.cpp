// flatten.cpp -- turns the pointer-shaped BVH of the reference
// (models/bvh.go:13-29: BVHNode with child pointers and a Bounds pointer;
// models/triangle.go:12-31: []*Triangle) into the two flat, 16-byte-aligned
// record arrays the kernels read (layout: rt_core.h, DevScene).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "rt_core.h"
#include "rt_internal.h"

namespace rt {

namespace {

// Records of the ordered traversal: a 4-wide tree collapsed from the
// reference's binary one.
//
// Step 1, small leaves: the reference's leaves hold up to BVHMaxLeafSize
// triangles; a leaf with more than kFastMaxLeaf is cut in two by array
// position (the range is sorted along its parent's split axis, bvh.go:73-76, so
// the parts are spatially separate; where to cut, and whether to cut a part of
// three or four triangles at all, is decided by surface area, see best_cut)
// under synthetic interior nodes with tight boxes, until every leaf is small: triangle tests are the poorly-vectorised
// part of the traversal.  Array positions, and with them hit ids and
// tie-breaking, are unchanged.
//
// Step 2, wide nodes: the traversal is bound by the number of dependent record
// fetches and of 16-byte loads per ray, so each 64-byte record carries FOUR
// child boxes: starting from a node's two children, the interior child whose
// surface area exceeds half the summed area of its own two children by most is
// replaced by those two (in place, so
// the reference's left-to-right order of the children is kept) until there are
// four or only leaves are left.  The boxes are stored as 8-bit offsets from
// the record's own frame (origin = lower corner of the union, one
// step per axis); lower planes are rounded down and upper planes up, and each
// is checked with the very fused multiply-add the traversal decodes it with,
// so a decoded box always CONTAINS the binary tree's box: the wide test passes
// whenever the reference's box test does.  Empty leaves (the reference can
// create one, bvh.go:94-100 with BVHMaxLeafSize 1) are dropped; unused slots
// carry the reference kLeafBit (no triangles) and are never entered.
struct WideBuilder {
    struct BNode {            // binary node after step 1
        float mn[3], mx[3];
        int32_t left = -1, right = -1;   // indices into `bin`, -1 for a leaf
        int32_t start = 0, count = 0;    // leaf: triangle range
    };
    const rt_mesh &mesh;
    const Bvh &bvh;
    const std::vector<int32_t> &skip;
    std::vector<float> &recs;
    std::vector<BNode> bin;
    int32_t max_wide_depth = 0;
    // Every decoded plane keeps this distance, per axis, from the box it bounds (lower
    // planes below, upper planes above).  The near-origin form of the traversal step
    // (rt_core.h: kNearOriginDiagonals) subtracts the ray origin from the record's frame
    // origin BEFORE it decodes a plane: d = fl(origin - o), plane - o = fma(offset, step,
    // d).  The one rounding of d errs by at most 2^-24 |origin - o|, so a plane that
    // lies at least that far outside the true box still yields a slab distance on the
    // safe side of the reference's (bound - o) * invDir (rounding is monotone).  The
    // margin is 2^-23 times the largest |origin - o| a qualifying ray can have.
    float margin[3] = {0, 0, 0};

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
    // Where to cut a leaf range of `count` triangles in two (0: leave it whole): the
    // first cut of the cheapest tree of cuts by position, when a leaf of this view
    // -- at most kFastMaxLeaf triangles -- costs kLeafCost times its box area per
    // pair of triangles, and every part that is cut its own area plus its two
    // parts.  A leaf visit costs several box tests: it takes the wave through a
    // change of phase at half-empty lanes.  Parts of one or two triangles are
    // never cut.
    static constexpr int32_t kCutSearch = 16;
    static constexpr float kLeafCost = 4.0f;
    int32_t best_cut(int32_t start, int32_t count) const {
        float cost[kCutSearch + 1][kCutSearch + 1];   // [first][length]
        int32_t cut[kCutSearch + 1][kCutSearch + 1];
        // box area of every sub-range [i, i + len): each triangle's own box once, then
        // unions grown one triangle at a time (min and max do not depend on the order)
        float area_of[kCutSearch + 1][kCutSearch + 1];
        {
            float tmn[kCutSearch][3], tmx[kCutSearch][3];
            for (int32_t i = 0; i < count; ++i) range_box(start + i, 1, tmn[i], tmx[i]);
            for (int32_t i = 0; i < count; ++i) {
                float mn[3] = {tmn[i][0], tmn[i][1], tmn[i][2]}, mx[3] = {tmx[i][0], tmx[i][1], tmx[i][2]};
                for (int32_t len = 1; i + len <= count; ++len) {
                    const int32_t j = i + len - 1;
                    for (int k = 0; k < 3; ++k) {
                        if (tmn[j][k] < mn[k]) mn[k] = tmn[j][k];
                        if (tmx[j][k] > mx[k]) mx[k] = tmx[j][k];
                    }
                    const float d0 = mx[0] - mn[0], d1 = mx[1] - mn[1], d2 = mx[2] - mn[2];
                    area_of[i][len] = d0 * d1 + d0 * d2 + d1 * d2;
                }
            }
        }
        for (int32_t len = 1; len <= count; ++len)
            for (int32_t i = 0; i + len <= count; ++i) {
                const float a = area_of[i][len];
                const float whole = a * kLeafCost * (float)((len + 1) / 2);
                cut[i][len] = 0;
                cost[i][len] = whole;
                if (len <= rtc::kFastSplitLeaf) continue;
                float best = std::numeric_limits<float>::max();
                int32_t at = len / 2;
                for (int32_t k = 1; k < len; ++k) {
                    const float c = cost[i][k] + cost[i + k][len - k];
                    if (c < best) {
                        best = c;
                        at = k;
                    }
                }
                if (len > rtc::kFastMaxLeaf || a + best <= whole) {
                    cost[i][len] = a + best;
                    cut[i][len] = at;
                }
            }
        return cut[0][count];
    }
    // step 1 for one leaf range whose box is already known
    int32_t small_leaf(int32_t start, int32_t count, const float mn[3], const float mx[3]) {
        const int32_t id = (int32_t)bin.size();
        bin.emplace_back();
        std::memcpy(bin[(size_t)id].mn, mn, 12);
        std::memcpy(bin[(size_t)id].mx, mx, 12);
        const int32_t half = count <= rtc::kFastSplitLeaf ? 0 : (count <= kCutSearch ? best_cut(start, count) : count / 2);
        if (half == 0) {
            bin[(size_t)id].start = start;
            bin[(size_t)id].count = count;
            return id;
        }
        float lmn[3], lmx[3];
        range_box(start, half, lmn, lmx);
        const int32_t l = small_leaf(start, half, lmn, lmx);
        range_box(start + half, count - half, lmn, lmx);
        const int32_t r = small_leaf(start + half, count - half, lmn, lmx);
        bin[(size_t)id].left = l;
        bin[(size_t)id].right = r;
        return id;
    }
    // binary tree of the reference (preorder, skip links) -> `bin`, iteratively
    int32_t import_tree() {
        const int32_t nn = bvh.n_nodes;
        std::vector<int32_t> id_of((size_t)nn, -1);
        // parents before children is not needed: create nodes in reverse preorder so
        // that both children exist when a parent is made
        for (int32_t i = nn - 1; i >= 0; --i) {
            const float *mn = &bvh.bmin[3 * (size_t)i], *mx = &bvh.bmax[3 * (size_t)i];
            if (bvh.is_leaf[(size_t)i]) {
                const int32_t start = bvh.start[(size_t)i];
                const int32_t count = bvh.end[(size_t)i] - start + 1;
                if (count <= 0) {
                    const float nan_ = std::numeric_limits<float>::quiet_NaN();
                    const float nn3[3] = {nan_, nan_, nan_};
                    id_of[(size_t)i] = small_leaf(0, 0, nn3, nn3);   // NaN bounds fail every comparison of the slab test
                } else {
                    id_of[(size_t)i] = small_leaf(start, count, mn, mx);
                }
            } else {
                const int32_t id = (int32_t)bin.size();
                bin.emplace_back();
                std::memcpy(bin[(size_t)id].mn, mn, 12);
                std::memcpy(bin[(size_t)id].mx, mx, 12);
                bin[(size_t)id].left = id_of[(size_t)i + 1];
                bin[(size_t)id].right = id_of[(size_t)skip[(size_t)i + 1]];
                id_of[(size_t)i] = id;
            }
        }
        return id_of[0];
    }
    static float area(const BNode &n) {
        const float d0 = n.mx[0] - n.mn[0], d1 = n.mx[1] - n.mn[1], d2 = n.mx[2] - n.mn[2];
        return d0 * d1 + d0 * d2 + d1 * d2;
    }
    uint32_t leaf_ref(const BNode &n) const {
        return rtc::kLeafBit | ((uint32_t)n.count << 24) | (uint32_t)n.start;
    }
    // step 2: the record of binary interior node `id`; returns its index
    uint32_t emit(int32_t id, int32_t depth) {
        if (depth > max_wide_depth) max_wide_depth = depth;
        const uint32_t rec = (uint32_t)(recs.size() / rtc::kWideFloats);
        recs.resize(recs.size() + rtc::kWideFloats, 0.0f);
        int32_t kids[4] = {bin[(size_t)id].left, bin[(size_t)id].right, -1, -1};
        int n = 2;
        while (n < 4) {
            int best = -1;
            float best_area = -1.0f;
            for (int i = 0; i < n; ++i) {
                const BNode &c = bin[(size_t)kids[i]];
                if (c.left < 0) continue;
                // what opening c saves: its own box no longer stands between the ray and its
                // children, whose boxes then cost a test each (weight 1/2: measured optimum)
                float a = area(c) - 0.5f * (area(bin[(size_t)c.left]) + area(bin[(size_t)c.right]));
                if (!(a == a)) a = -1.0f;          // a NaN box (empty subtree) is opened last
                if (best < 0 || a > best_area) {
                    best = i;
                    best_area = a;
                }
            }
            if (best < 0) break;
            const int32_t open = kids[best];
            for (int i = n; i > best + 1; --i) kids[i] = kids[i - 1];
            kids[best] = bin[(size_t)open].left;
            kids[best + 1] = bin[(size_t)open].right;
            ++n;
        }
        // drop empty leaves (NaN boxes)
        {
            int m = 0;
            for (int i = 0; i < n; ++i) {
                const BNode &c = bin[(size_t)kids[i]];
                if (c.left < 0 && c.count <= 0) continue;
                kids[m++] = kids[i];
            }
            n = m;
        }
        // frame: lower corner of the union and a step per axis
        float origin[3], step[3];
        uint8_t qlo[4][3], qhi[4][3];
        for (int a = 0; a < 3; ++a) {
            float lo = std::numeric_limits<float>::max(), hi = -std::numeric_limits<float>::max();
            for (int i = 0; i < n; ++i) {
                const BNode &c = bin[(size_t)kids[i]];
                if (c.mn[a] < lo) lo = c.mn[a];
                if (c.mx[a] > hi) hi = c.mx[a];
            }
            if (n == 0) { lo = 0; hi = 0; }
            // the frame starts a margin below the lowest child plane, rounded down
            const double m = (double)margin[a];
            float fo = (float)((double)lo - m);
            if ((double)fo > (double)lo - m) fo = std::nextafter(fo, -std::numeric_limits<float>::infinity());
            lo = fo;
            origin[a] = lo;
            const float ext = (float)(((double)hi + m) - (double)lo) * 1.0000002f;
            // smallest step s with 255 * s >= ext that the check below accepts (any float, not only a
            // power of two: the decode is one fma either way); a floor keeps empty extents finite
            float s = ext > 0 ? ext / 255.0f : 0.0f;
            if (!(s > 7.5e-37f)) s = 7.5e-37f;
            for (;; s *= 1.000001f) {
                bool ok = true;
                for (int i = 0; i < n && ok; ++i) {
                    const BNode &c = bin[(size_t)kids[i]];
                    // A plane at offset q lies at q * s + lo exactly (a product of two floats plus
                    // a float is exact in double arithmetic for these magnitudes); the decode
                    // fma(q, s, lo) rounds that value to a float, and rounding never crosses the
                    // box's own float bound, so a plane that is outside by the margin in exact
                    // arithmetic also decodes outward.
                    auto exact = [&](int q) { return (double)q * (double)s + (double)lo; };
                    // lower plane: largest q whose plane lies at least the margin below the bound
                    const double want_lo = (double)c.mn[a] - m;
                    float x = std::floor((float)((want_lo - (double)lo) / (double)s));
                    if (!(x >= 0)) x = 0;
                    if (x > 255) x = 255;
                    int q = (int)x;
                    while (q > 0 && exact(q) > want_lo) --q;
                    if (exact(q) > want_lo || std::fmaf((float)q, s, lo) > c.mn[a]) ok = false;
                    qlo[i][a] = (uint8_t)q;
                    // upper plane: smallest q whose plane lies at least the margin above the bound
                    const double want_hi = (double)c.mx[a] + m;
                    x = std::ceil((float)((want_hi - (double)lo) / (double)s));
                    if (!(x >= 0)) x = 0;
                    if (x > 255) { ok = false; x = 255; }
                    q = (int)x;
                    while (q < 255 && exact(q) < want_hi) ++q;
                    if (exact(q) < want_hi || std::fmaf((float)q, s, lo) < c.mx[a]) ok = false;
                    qhi[i][a] = (uint8_t)q;
                }
                if (ok || !std::isfinite(s)) { step[a] = s; break; }
            }
        }
        // An unused slot carries an inverted box: lower planes at offset 255, upper
        // planes at offset 0.  On an axis where offset 255 decodes above offset 0 the
        // slab interval of such a box is empty for every ray, so the traversal needs
        // no "slot in use" test.  A record whose children are all one point has no
        // such axis yet: widen its x step until it has one (its children sit at
        // offset 0 on that axis and stay where they are).
        if (n < 4) {
            bool strict_axis = false;
            for (int a = 0; a < 3; ++a)
                if (std::fmaf(255.0f, step[a], origin[a]) > origin[a]) strict_axis = true;
            if (!strict_axis) {
                float s = std::fabs(origin[0]) * 1e-9f;
                if (!(s > step[0])) s = step[0];
                while (!(std::fmaf(255.0f, s, origin[0]) > origin[0]) && std::isfinite(s)) s *= 2.0f;
                step[0] = s;
            }
        }
        float *p = &recs[(size_t)rec * rtc::kWideFloats];
        p[0] = origin[0]; p[1] = origin[1]; p[2] = origin[2];
        p[3] = step[0]; p[4] = step[1]; p[5] = step[2];
        uint32_t planes[6] = {0, 0, 0, 0, 0, 0};   // lo.x lo.y lo.z hi.x hi.y hi.z, child c in byte c
        for (int i = 0; i < 4; ++i)
            for (int a = 0; a < 3; ++a) {
                planes[a] |= (uint32_t)(i < n ? qlo[i][a] : 255) << (8 * i);
                planes[3 + a] |= (uint32_t)(i < n ? qhi[i][a] : 0) << (8 * i);
            }
        for (int j = 0; j < 6; ++j) p[6 + j] = rtc::i2f((int32_t)planes[j]);
        for (int i = 0; i < 4; ++i) {
            uint32_t ref = rtc::kLeafBit;            // unused slot: a leaf of no triangles behind an inverted box
            if (i < n) {
                const BNode c = bin[(size_t)kids[i]];
                ref = c.left < 0 ? leaf_ref(c) : emit(kids[i], depth + 1);
            }
            recs[(size_t)rec * rtc::kWideFloats + 12 + (size_t)i] = rtc::i2f((int32_t)ref);
        }
        return rec;
    }
};

// Moves the kWideTopRecords records with the largest frames to the front of the
// array, largest first, and leaves the others in the order they were emitted in
// (depth first, a record's first interior child right behind it).  A record's
// frame contains its children's frames, so the front of the array is a top of the
// tree, and a frame's surface area is how likely a ray is to reach it: a traversal
// kernel that keeps the first K records in LDS, for any K up to that constant,
// serves most record fetches from there (on a 1M-triangle room 64 records serve
// about half of them, 512 three quarters).
void top_records_first(std::vector<float> &recs, uint32_t &root) {
    const size_t n = recs.size() / rtc::kWideFloats;
    if (n < 2) return;
    const size_t n_top = n < (size_t)rtc::kWideTopRecords ? n : (size_t)rtc::kWideTopRecords;
    std::vector<double> area(n);
    for (size_t i = 0; i < n; ++i) {
        const float *p = &recs[i * rtc::kWideFloats];
        const double x = 255.0 * p[3], y = 255.0 * p[4], z = 255.0 * p[5];
        const double a = x * y + x * z + y * z;
        area[i] = a == a ? a : -1.0;
    }
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    auto larger = [&](uint32_t x, uint32_t y) { return area[x] != area[y] ? area[x] > area[y] : x < y; };
    std::partial_sort(order.begin(), order.begin() + (std::ptrdiff_t)n_top, order.end(), larger);
    std::sort(order.begin() + (std::ptrdiff_t)n_top, order.end());
    std::vector<uint32_t> new_of(n);
    for (size_t k = 0; k < n; ++k) new_of[order[k]] = (uint32_t)k;
    std::vector<float> moved(recs.size());
    for (size_t k = 0; k < n; ++k) {
        float *q = &moved[k * rtc::kWideFloats];
        std::memcpy(q, &recs[(size_t)order[k] * rtc::kWideFloats], sizeof(float) * rtc::kWideFloats);
        for (int c = 0; c < 4; ++c) {
            const uint32_t ref = (uint32_t)rtc::f2i(q[12 + c]);
            if (!(ref & rtc::kLeafBit)) q[12 + c] = rtc::i2f((int32_t)new_of[ref]);
        }
    }
    recs.swap(moved);
    if (!(root & rtc::kLeafBit)) root = new_of[root];
}

void build_wide(const rt_mesh &mesh, const Bvh &bvh, const std::vector<int32_t> &skip, FlatScene &out) {
    out.wide.clear();
    out.fast_ok = false;
    out.fast_root = 0;
    if (mesh.n_tris > rtc::kFastMaxTris || bvh.max_depth > 60) return;
    WideBuilder b{mesh, bvh, skip, out.wide, {}, 0, {0, 0, 0}};
    b.bin.reserve((size_t)bvh.n_nodes * 2);
    {
        // rays of the near-origin form start within kNearOriginDiagonals scene diagonals of
        // the scene's box (the root's), a record's frame origin lies inside that box: on
        // every axis |origin - o| is at most the box's extent plus that many diagonals
        const float *mn = &bvh.bmin[0], *mx = &bvh.bmax[0];
        double diag2 = 0;
        for (int a = 0; a < 3; ++a) diag2 += ((double)mx[a] - mn[a]) * ((double)mx[a] - mn[a]);
        const double diag = std::sqrt(diag2);
        for (int a = 0; a < 3; ++a) {
            const double reach = ((double)mx[a] - mn[a]) + (double)rtc::kNearOriginDiagonals * diag;
            const double m = std::isfinite(reach) ? reach * 0x1p-23 : 0.0;
            b.margin[a] = (float)m;
            out.near_lo[a] = (float)((double)mn[a] - (double)rtc::kNearOriginDiagonals * diag);
            out.near_hi[a] = (float)((double)mx[a] + (double)rtc::kNearOriginDiagonals * diag);
        }
    }
    const int32_t root = b.import_tree();
    const WideBuilder::BNode &rn = b.bin[(size_t)root];
    if (rn.left < 0) {
        if (rn.count <= 0) return;      // a tree whose root is an empty leaf has nothing to traverse in order
        out.fast_root = b.leaf_ref(rn); // the root is entered without a box test
        out.wide.resize(rtc::kWideFloats, 0.0f);
    } else {
        out.fast_root = b.emit(root, 1);
    }
    // every level can leave three entries on a lane's stack
    if (3 * b.max_wide_depth > rtc::kFastStackEntries) {
        out.wide.clear();
        out.fast_root = 0;
        return;
    }
    if (rn.left >= 0) top_records_first(out.wide, out.fast_root);
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
    }
    // the per-triangle records are independent of each other: large meshes fill them in
    // slices on several threads
    auto fill = [&](int32_t first, int32_t last) {
        for (int32_t pos = first; pos < last; ++pos) {
            const int32_t id = bvh.order[(size_t)pos];
            TriRecord rec;
            make_tri_record(mesh.verts + 9 * (size_t)id, rec);
            tris[3 * (size_t)pos] = f4{rec.normal[0], rec.normal[1], rec.normal[2], rec.v0[0]};
            tris[3 * (size_t)pos + 1] = f4{rec.v0[1], rec.v0[2], rec.e0[0], rec.e0[1]};
            tris[3 * (size_t)pos + 2] = f4{rec.e0[2], rec.v0v2[0], rec.v0v2[1], rec.v0v2[2]};
            const int32_t mat = mesh.material ? mesh.material[id] : 0;
            out.tri_mat[(size_t)pos] = mat | (mesh.is_light[mat] ? rtc::kLightBit : 0);
            if (mesh.texcoords) std::memcpy(&out.tri_uv[6 * (size_t)pos], mesh.texcoords + 6 * (size_t)id, 24);
        }
    };
    if (mesh.n_tris > 200000) {
        const int32_t slices = 8;
        std::vector<std::function<void()>> jobs;
        for (int32_t k = 0; k < slices; ++k) {
            const int32_t first = (int32_t)((int64_t)mesh.n_tris * k / slices), last = (int32_t)((int64_t)mesh.n_tris * (k + 1) / slices);
            jobs.push_back([&fill, first, last] { fill(first, last); });
        }
        run_together(jobs);
    } else {
        fill(0, mesh.n_tris);
    }
    build_wide(mesh, bvh, skip, out);
    return RT_OK;
}

}  // namespace rt
