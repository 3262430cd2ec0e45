// bvh_io.cpp -- BVH hand-over formats.
//   JSON: the text main.go:65-76 (buildBVH) returns and main.go:78-90
//         (loadBVH) accepts: encoding/json of models.BVH / BVHNode / AABB
//         (bvh.go:13-29, aabb.go:11-22).  Loading re-sorts the triangle array
//         node by node from the stored split planes, as BVHNode.load does
//         (bvh.go:44-54).
//   binary: the same preorder arrays in a flat file, for the on-disk cache
//         (the reference keeps the JSON in the browser's IndexedDB,
//         frontend Renderer.js:235-274).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>

#include "json.h"
#include "rt_internal.h"

namespace rt {

namespace {

void append_vec(std::string &s, const float *v, int n) {
    s.push_back('[');
    for (int i = 0; i < n; ++i) {
        if (i) s.push_back(',');
        rtjson::append_f32(s, v[i]);
    }
    s.push_back(']');
}

// subtree sizes from the preorder leaf flags
std::vector<int32_t> subtree_sizes(const Bvh &b) {
    std::vector<int32_t> size((size_t)b.n_nodes, 1);
    for (int32_t i = b.n_nodes - 1; i >= 0; --i) {
        if (!b.is_leaf[(size_t)i]) {
            int32_t l = i + 1;
            int32_t r = l + size[(size_t)l];
            size[(size_t)i] = 1 + size[(size_t)l] + (r < b.n_nodes ? size[(size_t)r] : 0);
        }
    }
    return size;
}

void write_node(const Bvh &b, const std::vector<int32_t> &size, int32_t i, std::string &s) {
    // iterative emission: a stack of (node, phase) keeps deep trees off the call stack
    struct Frame {
        int32_t node;
        int phase;
    };
    std::vector<Frame> st;
    st.push_back({i, 0});
    while (!st.empty()) {
        Frame &f = st.back();
        const int32_t n = f.node;
        const bool leaf = b.is_leaf[(size_t)n] != 0;
        if (f.phase == 0) {
            s += "{\"Depth\":" + std::to_string(b.depth[(size_t)n]) + ",\"Index\":" + std::to_string(n) + ",\"LeftChild\":";
            f.phase = 1;
            if (leaf) {
                s += "null";
            } else {
                st.push_back({n + 1, 0});
            }
            continue;
        }
        if (f.phase == 1) {
            s += ",\"RightChild\":";
            f.phase = 2;
            if (leaf) {
                s += "null";
            } else {
                st.push_back({n + 1 + size[(size_t)n + 1], 0});
            }
            continue;
        }
        const float *mn = &b.bmin[3 * (size_t)n], *mx = &b.bmax[3 * (size_t)n];
        // NewAABBMinMax (aabb.go:46-56)
        float pos[3] = {(mn[0] + mx[0]) * 0.5f, (mn[1] + mx[1]) * 0.5f, (mn[2] + mx[2]) * 0.5f};
        s += ",\"Bounds\":{\"Position\":";
        append_vec(s, pos, 3);
        s += ",\"Width\":";
        rtjson::append_f32(s, mx[0] - mn[0]);
        s += ",\"Height\":";
        rtjson::append_f32(s, mx[1] - mn[1]);
        s += ",\"Length\":";
        rtjson::append_f32(s, mx[2] - mn[2]);
        s += ",\"Min\":";
        append_vec(s, mn, 3);
        s += ",\"Max\":";
        append_vec(s, mx, 3);
        s += ",\"Bounds\":[";
        append_vec(s, mn, 3);
        s.push_back(',');
        append_vec(s, mx, 3);
        s += "]},\"StartIndex\":" + std::to_string(b.start[(size_t)n]) + ",\"EndIndex\":" + std::to_string(b.end[(size_t)n]) + ",\"SplitPlane\":";
        append_vec(s, &b.split[4 * (size_t)n], 4);
        s.push_back('}');
        st.pop_back();
    }
}

bool read_vec(const rtjson::Value *v, float *out, int n) {
    if (!v || v->kind != rtjson::Value::Array || (int)v->arr.size() != n) return false;
    for (int i = 0; i < n; ++i) {
        if (v->arr[(size_t)i].kind != rtjson::Value::Number) return false;
        out[i] = (float)v->arr[(size_t)i].num;
    }
    return true;
}

}  // namespace

rt_status bvh_to_json(const Bvh &b, std::string &out) {
    out.clear();
    out.reserve((size_t)b.n_nodes * 400 + 64);
    out += "{\"Root\":";
    if (b.n_nodes == 0) {
        out += "null";
    } else {
        write_node(b, subtree_sizes(b), 0, out);
    }
    out += "}";
    return RT_OK;
}

// TriangleSorter over order[start..end] (triangle.go:67-76) for an arbitrary
// axis vector; ties on (key, Index) fall back to the input position.
static void sort_range(const rt_mesh &mesh, std::vector<int32_t> &order, int32_t start,
                       int32_t end, const float axis[3]) {
    if (end <= start) return;
    struct Item {
        float key;
        int32_t index, id;
    };
    const float third = (float)(1.0 / 3.0);
    std::vector<Item> items((size_t)(end - start + 1));
    for (int32_t i = start; i <= end; ++i) {
        const int32_t id = order[(size_t)i];
        const float *v = mesh.verts + 9 * (size_t)id;
        float c[3];
        for (int k = 0; k < 3; ++k) c[k] = ((v[k] + v[3 + k]) + v[6 + k]) * third;
        items[(size_t)(i - start)] = {axis[0] * c[0] + axis[1] * c[1] + axis[2] * c[2],
                                      mesh.group_index ? mesh.group_index[id] : id, id};
    }
    std::sort(items.begin(), items.end(), [](const Item &a, const Item &b) {
        if (a.key == b.key) {
            if (a.index != b.index) return a.index < b.index;
            return a.id < b.id;
        }
        return a.key < b.key;
    });
    for (int32_t i = start; i <= end; ++i) order[(size_t)i] = items[(size_t)(i - start)].id;
}

// BVH.Load: sort every interior node's range by its stored plane, parents
// before children.
static rt_status resort_from_planes(const rt_mesh &mesh, Bvh &b) {
    b.order.resize((size_t)mesh.n_tris);
    for (int32_t i = 0; i < mesh.n_tris; ++i) b.order[(size_t)i] = i;
    for (int32_t n = 0; n < b.n_nodes; ++n) {
        const int32_t s = b.start[(size_t)n], e = b.end[(size_t)n];
        if (e >= mesh.n_tris || s < 0)
            return fail(RT_ERR_INVALID, "BVH node " + std::to_string(n) + " indexes outside the triangle array");
        if (!b.is_leaf[(size_t)n]) sort_range(mesh, b.order, s, e, &b.split[4 * (size_t)n]);
    }
    return RT_OK;
}

rt_status bvh_from_json(const rt_mesh &mesh, const char *json, size_t len, Bvh &out) {
    rtjson::Value doc;
    std::string err;
    if (!rtjson::parse(json, len, doc, err)) return fail(RT_ERR_PARSE, "BVH JSON: " + err);
    const rtjson::Value *root = doc.get("Root");
    if (!root || root->kind != rtjson::Value::Object) return fail(RT_ERR_PARSE, "BVH JSON has no Root object");
    out = Bvh();
    out.n_tris = mesh.n_tris;
    std::vector<std::pair<const rtjson::Value *, int32_t>> st;   // preorder walk, depth recomputed
    st.emplace_back(root, 0);
    while (!st.empty()) {
        const rtjson::Value *n = st.back().first;
        const int32_t depth = st.back().second;
        st.pop_back();
        const rtjson::Value *l = n->get("LeftChild"), *r = n->get("RightChild");
        const bool has_l = l && l->kind == rtjson::Value::Object;
        const bool has_r = r && r->kind == rtjson::Value::Object;
        if (has_l != has_r) return fail(RT_ERR_PARSE, "BVH JSON node with a single child");
        float mn[3], mx[3], split[4] = {0, 0, 0, 0};
        const rtjson::Value *bounds = n->get("Bounds");
        bool ok = false;
        if (bounds && bounds->kind == rtjson::Value::Object) {
            const rtjson::Value *bb = bounds->get("Bounds");
            if (bb && bb->kind == rtjson::Value::Array && bb->arr.size() == 2)
                ok = read_vec(&bb->arr[0], mn, 3) && read_vec(&bb->arr[1], mx, 3);
            if (!ok) ok = read_vec(bounds->get("Min"), mn, 3) && read_vec(bounds->get("Max"), mx, 3);
        }
        if (!ok) return fail(RT_ERR_PARSE, "BVH JSON node without usable Bounds");
        read_vec(n->get("SplitPlane"), split, 4);
        out.is_leaf.push_back(has_l ? 0 : 1);
        const double si = n->number_or("StartIndex", 0), ei = n->number_or("EndIndex", -1);
        if (!(si >= -1 && si <= 2147483647.0 && ei >= -1 && ei <= 2147483647.0))
            return fail(RT_ERR_PARSE, "BVH JSON node with an index outside the int range");
        out.start.push_back((int32_t)si);
        out.end.push_back((int32_t)ei);
        out.depth.push_back(depth);
        out.max_depth = std::max(out.max_depth, depth);
        for (int k = 0; k < 3; ++k) {
            out.bmin.push_back(mn[k]);
            out.bmax.push_back(mx[k]);
        }
        for (int k = 0; k < 4; ++k) out.split.push_back(split[k]);
        if (has_l) {
            st.emplace_back(r, depth + 1);
            st.emplace_back(l, depth + 1);
        }
    }
    out.n_nodes = (int32_t)out.is_leaf.size();
    return resort_from_planes(mesh, out);
}

namespace {
constexpr uint64_t kMagic = 0x3130485642544152ULL;   // "RATBVH01"
struct FileHeader {
    uint64_t magic;
    int32_t n_tris, n_nodes, max_depth, reserved;
};
template <typename T>
bool write_vec(std::FILE *f, const std::vector<T> &v) {
    return v.empty() || std::fwrite(v.data(), sizeof(T), v.size(), f) == v.size();
}
template <typename T>
bool read_vec_file(std::FILE *f, std::vector<T> &v, size_t n) {
    v.resize(n);
    return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n;
}
}  // namespace

rt_status bvh_save(const Bvh &b, const char *path) {
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return fail(RT_ERR_INVALID, std::string("cannot open ") + path + " for writing");
    FileHeader h{kMagic, b.n_tris, b.n_nodes, b.max_depth, 0};
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && write_vec(f, b.is_leaf) && write_vec(f, b.start) &&
              write_vec(f, b.end) && write_vec(f, b.depth) && write_vec(f, b.bmin) &&
              write_vec(f, b.bmax) && write_vec(f, b.split) && write_vec(f, b.order);
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) return fail(RT_ERR_INVALID, std::string("short write to ") + path);
    return RT_OK;
}

rt_status bvh_load(const rt_mesh &mesh, const char *path, Bvh &out) {
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return fail(RT_ERR_INVALID, std::string("cannot open ") + path);
    FileHeader h{};
    out = Bvh();
    bool ok = std::fread(&h, sizeof h, 1, f) == 1 && h.magic == kMagic && h.n_nodes >= 0 && h.n_tris >= 0;
    if (ok && h.n_tris != mesh.n_tris) {
        std::fclose(f);
        return fail(RT_ERR_INVALID, "cached BVH was built for a different triangle count");
    }
    const size_t nn = ok ? (size_t)h.n_nodes : 0;
    ok = ok && read_vec_file(f, out.is_leaf, nn) && read_vec_file(f, out.start, nn) &&
         read_vec_file(f, out.end, nn) && read_vec_file(f, out.depth, nn) &&
         read_vec_file(f, out.bmin, nn * 3) && read_vec_file(f, out.bmax, nn * 3) &&
         read_vec_file(f, out.split, nn * 4) && read_vec_file(f, out.order, (size_t)h.n_tris);
    std::fclose(f);
    if (!ok) return fail(RT_ERR_PARSE, std::string("not a BVH cache file: ") + path);
    out.n_tris = h.n_tris;
    out.n_nodes = h.n_nodes;
    out.max_depth = h.max_depth;
    for (int32_t id : out.order)
        if (id < 0 || id >= h.n_tris) return fail(RT_ERR_PARSE, "BVH cache holds an invalid triangle id");
    return RT_OK;
}

}  // namespace rt
