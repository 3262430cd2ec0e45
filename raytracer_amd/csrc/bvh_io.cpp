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
#include <atomic>
#include <cstdio>
#include <cstring>
#include <functional>
#include <thread>

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
// before children.  A child's range lies inside its parent's and the ranges of
// nodes at one depth do not overlap (checked below; a tree that breaks this is
// sorted node by node in preorder instead), so the nodes of one depth are
// sorted on several threads, depth by depth.
static rt_status resort_from_planes(const rt_mesh &mesh, Bvh &b) {
    b.order.resize((size_t)mesh.n_tris);
    for (int32_t i = 0; i < mesh.n_tris; ++i) b.order[(size_t)i] = i;
    std::vector<std::vector<int32_t>> by_depth;
    for (int32_t n = 0; n < b.n_nodes; ++n) {
        const int32_t s = b.start[(size_t)n], e = b.end[(size_t)n];
        if (e >= mesh.n_tris || s < 0)
            return fail(RT_ERR_INVALID, "BVH node " + std::to_string(n) + " indexes outside the triangle array");
        if (b.is_leaf[(size_t)n]) continue;
        const size_t d = (size_t)b.depth[(size_t)n];
        if (d >= by_depth.size()) by_depth.resize(d + 1);
        by_depth[d].push_back(n);
    }
    bool disjoint = true;   // preorder within a depth is left to right: ranges must ascend
    for (const auto &level : by_depth)
        for (size_t i = 1; i < level.size() && disjoint; ++i)
            if (b.start[(size_t)level[i]] <= b.end[(size_t)level[i - 1]]) disjoint = false;
    const unsigned hw = std::thread::hardware_concurrency();
    const int n_threads = (int)(hw == 0 ? 1 : (hw > 16 ? 16 : hw));
    if (!disjoint || n_threads < 2 || mesh.n_tris < 50000) {
        for (int32_t n = 0; n < b.n_nodes; ++n)
            if (!b.is_leaf[(size_t)n])
                sort_range(mesh, b.order, b.start[(size_t)n], b.end[(size_t)n], &b.split[4 * (size_t)n]);
        return RT_OK;
    }
    for (const auto &level : by_depth) {
        if (level.size() < 2) {
            for (int32_t n : level)
                sort_range(mesh, b.order, b.start[(size_t)n], b.end[(size_t)n], &b.split[4 * (size_t)n]);
            continue;
        }
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t i = next.fetch_add(1); i < level.size(); i = next.fetch_add(1)) {
                const int32_t n = level[i];
                sort_range(mesh, b.order, b.start[(size_t)n], b.end[(size_t)n], &b.split[4 * (size_t)n]);
            }
        };
        const int use = (int)std::min<size_t>((size_t)n_threads, level.size());
        run_together(std::vector<std::function<void()>>((size_t)use, work));
    }
    return RT_OK;
}

namespace {

// One BVHNode of the hand-over text, as read: the tree of a million triangles
// is a 110 MB document, so it is read token by token into these 64-byte
// structs instead of a DOM.
struct TextNode {
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0}, split[4] = {0, 0, 0, 0};
    int32_t start = 0, end = -1, left = -1, right = -1;
    bool have_bounds = false;
};

class TreeReader {
  public:
    TreeReader(const char *s, size_t n) : lx(s, n) {}
    rtjson::Lexer lx;
    std::vector<TextNode> nodes;
    std::string problem;   // set when the text is valid JSON but not a usable tree

    // `n` numbers in an array; anything else is stepped over and reported as unusable
    bool numbers(float *out, int n, bool &usable) {
        usable = false;
        if (lx.peek() != '[') return lx.skip_value();
        lx.consume('[');
        int count = 0;
        bool bad = false;
        if (!lx.consume(']')) {
            for (;;) {
                const char c = lx.peek();
                if (c == '-' || (c >= '0' && c <= '9')) {
                    double d;
                    if (!lx.number(d)) return false;
                    if (count < n) out[count] = (float)d;
                    ++count;
                } else {
                    bad = true;
                    if (!lx.skip_value()) return false;
                }
                if (lx.consume(',')) continue;
                if (lx.consume(']')) break;
                return lx.fail("expected ',' or ']'");
            }
        }
        usable = !bad && count == n;
        return true;
    }
    bool index(int32_t &out) {
        const char c = lx.peek();
        if (!(c == '-' || (c >= '0' && c <= '9'))) return lx.skip_value();   // wrong type: keeps the default
        double d;
        if (!lx.number(d)) return false;
        if (!(d >= -1 && d <= 2147483647.0)) {
            if (problem.empty()) problem = "BVH JSON node with an index outside the int range";
            return true;
        }
        out = (int32_t)d;
        return true;
    }
    // models.AABB: "Bounds":[min,max] first, else "Min" and "Max"
    bool bounds(TextNode &nd) {
        if (lx.peek() != '{') return lx.skip_value();
        lx.consume('{');
        float pair_mn[3], pair_mx[3], mn[3], mx[3];
        bool pair_ok = false, mn_ok = false, mx_ok = false;
        std::string key;
        if (!lx.consume('}')) {
            for (;;) {
                if (!lx.string(key)) return false;
                if (!lx.consume(':')) return lx.fail("expected ':'");
                if (rtjson::key_is(key, "Bounds")) {
                    pair_ok = false;
                    if (lx.peek() != '[') {
                        if (!lx.skip_value()) return false;
                    } else {
                        lx.consume('[');
                        int count = 0;
                        bool a = false, b = false, extra = false;
                        if (!lx.consume(']')) {
                            for (;;) {
                                bool ok = false;
                                if (count == 0) {
                                    if (!numbers(pair_mn, 3, ok)) return false;
                                    a = ok;
                                } else if (count == 1) {
                                    if (!numbers(pair_mx, 3, ok)) return false;
                                    b = ok;
                                } else {
                                    extra = true;
                                    if (!lx.skip_value()) return false;
                                }
                                ++count;
                                if (lx.consume(',')) continue;
                                if (lx.consume(']')) break;
                                return lx.fail("expected ',' or ']'");
                            }
                        }
                        pair_ok = a && b && !extra;
                    }
                } else if (rtjson::key_is(key, "Min")) {
                    if (!numbers(mn, 3, mn_ok)) return false;
                } else if (rtjson::key_is(key, "Max")) {
                    if (!numbers(mx, 3, mx_ok)) return false;
                } else if (!lx.skip_value()) {
                    return false;
                }
                if (lx.consume(',')) continue;
                if (lx.consume('}')) break;
                return lx.fail("expected ',' or '}'");
            }
        }
        if (pair_ok) {
            std::memcpy(nd.mn, pair_mn, 12);
            std::memcpy(nd.mx, pair_mx, 12);
            nd.have_bounds = true;
        } else if (mn_ok && mx_ok) {
            std::memcpy(nd.mn, mn, 12);
            std::memcpy(nd.mx, mx, 12);
            nd.have_bounds = true;
        }
        return true;
    }
    // A BVHNode object; returns its index in `nodes`, -1 for anything that is not
    // an object (null, or a value of another type), -2 on a syntax error.
    int32_t node(int depth) {
        if (depth > 2000) {
            lx.fail("BVH JSON nested too deeply");
            return -2;
        }
        if (lx.peek() != '{') return lx.skip_value() ? -1 : -2;
        lx.consume('{');
        const int32_t id = (int32_t)nodes.size();
        nodes.emplace_back();
        std::string key;
        if (lx.consume('}')) return id;
        for (;;) {
            if (!lx.string(key)) return -2;
            if (!lx.consume(':')) {
                lx.fail("expected ':'");
                return -2;
            }
            if (rtjson::key_is(key, "LeftChild") || rtjson::key_is(key, "RightChild")) {
                const int32_t child = node(depth + 1);
                if (child == -2) return -2;
                if (rtjson::key_is(key, "LeftChild")) nodes[(size_t)id].left = child;
                else nodes[(size_t)id].right = child;
            } else if (rtjson::key_is(key, "Bounds")) {
                TextNode nd = nodes[(size_t)id];
                if (!bounds(nd)) return -2;
                nodes[(size_t)id] = nd;
            } else if (rtjson::key_is(key, "StartIndex")) {
                if (!index(nodes[(size_t)id].start)) return -2;
            } else if (rtjson::key_is(key, "EndIndex")) {
                if (!index(nodes[(size_t)id].end)) return -2;
            } else if (rtjson::key_is(key, "SplitPlane")) {
                float sp[4] = {0, 0, 0, 0};
                bool ok = false;
                if (!numbers(sp, 4, ok)) return -2;
                if (ok) std::memcpy(nodes[(size_t)id].split, sp, 16);
            } else if (!lx.skip_value()) {
                return -2;
            }
            if (lx.consume(',')) continue;
            if (lx.consume('}')) return id;
            lx.fail("expected ',' or '}'");
            return -2;
        }
    }
    // the document: {"Root": node}
    bool document(int32_t &root) {
        root = -1;
        if (lx.peek() != '{') {
            if (!lx.skip_value()) return false;
            return lx.at_end() || lx.fail("trailing characters after JSON value");
        }
        lx.consume('{');
        std::string key;
        if (!lx.consume('}')) {
            for (;;) {
                if (!lx.string(key)) return false;
                if (!lx.consume(':')) return lx.fail("expected ':'");
                if (rtjson::key_is(key, "Root")) {
                    nodes.clear();
                    root = node(0);
                    if (root == -2) return false;
                } else if (!lx.skip_value()) {
                    return false;
                }
                if (lx.consume(',')) continue;
                if (lx.consume('}')) break;
                return lx.fail("expected ',' or '}'");
            }
        }
        return lx.at_end() || lx.fail("trailing characters after JSON value");
    }
};

}  // namespace

rt_status bvh_from_json(const rt_mesh &mesh, const char *json, size_t len, Bvh &out) {
    TreeReader reader(json, len);
    int32_t root = -1;
    if (!reader.document(root))
        return fail(RT_ERR_PARSE, "BVH JSON: " + (reader.lx.error().empty() ? std::string("invalid JSON") : reader.lx.error()));
    if (root < 0) return fail(RT_ERR_PARSE, "BVH JSON has no Root object");
    if (!reader.problem.empty()) return fail(RT_ERR_PARSE, reader.problem);
    const std::vector<TextNode> &tn = reader.nodes;
    out = Bvh();
    out.n_tris = mesh.n_tris;
    const size_t total = tn.size();
    out.is_leaf.reserve(total);
    out.start.reserve(total);
    out.end.reserve(total);
    out.depth.reserve(total);
    out.bmin.reserve(3 * total);
    out.bmax.reserve(3 * total);
    out.split.reserve(4 * total);
    std::vector<std::pair<int32_t, int32_t>> st;   // preorder walk, depth recomputed
    st.emplace_back(root, 0);
    while (!st.empty()) {
        const TextNode &n = tn[(size_t)st.back().first];
        const int32_t depth = st.back().second;
        st.pop_back();
        const bool has_l = n.left >= 0, has_r = n.right >= 0;
        if (has_l != has_r) return fail(RT_ERR_PARSE, "BVH JSON node with a single child");
        if (!n.have_bounds) return fail(RT_ERR_PARSE, "BVH JSON node without usable Bounds");
        out.is_leaf.push_back(has_l ? 0 : 1);
        out.start.push_back(n.start);
        out.end.push_back(n.end);
        out.depth.push_back(depth);
        out.max_depth = std::max(out.max_depth, depth);
        for (int k = 0; k < 3; ++k) {
            out.bmin.push_back(n.mn[k]);
            out.bmax.push_back(n.mx[k]);
        }
        for (int k = 0; k < 4; ++k) out.split.push_back(n.split[k]);
        if (has_l) {
            st.emplace_back(n.right, depth + 1);
            st.emplace_back(n.left, depth + 1);
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
    // Nothing is allocated from a header field before it is checked against the
    // file's real size: a binary tree over n triangles has at most 2n - 1 nodes
    // (one for an empty mesh), and the arrays that follow the header have a
    // fixed size per node and per triangle.
    if (ok) {
        const uint64_t nn64 = (uint64_t)h.n_nodes, nt64 = (uint64_t)h.n_tris;
        const uint64_t expect = sizeof(FileHeader) + nn64 * (4 * 4 + 3 * 4 + 3 * 4 + 4 * 4) + nt64 * 4;
        long size = -1;
        if (std::fseek(f, 0, SEEK_END) == 0) size = std::ftell(f);
        ok = size >= 0 && (uint64_t)size == expect && nn64 >= 1 && nn64 <= 2 * std::max<uint64_t>(nt64, 1) &&
             std::fseek(f, (long)sizeof(FileHeader), SEEK_SET) == 0;
    }
    const size_t nn = ok ? (size_t)h.n_nodes : 0;
    ok = ok && read_vec_file(f, out.is_leaf, nn) && read_vec_file(f, out.start, nn) &&
         read_vec_file(f, out.end, nn) && read_vec_file(f, out.depth, nn) &&
         read_vec_file(f, out.bmin, nn * 3) && read_vec_file(f, out.bmax, nn * 3) &&
         read_vec_file(f, out.split, nn * 4) && read_vec_file(f, out.order, (size_t)h.n_tris);
    std::fclose(f);
    if (!ok) return fail(RT_ERR_PARSE, std::string("not a BVH cache file (bad header, or its size does not match its node and triangle counts): ") + path);
    out.n_tris = h.n_tris;
    out.n_nodes = h.n_nodes;
    // the order must be a permutation of the triangle ids
    std::vector<uint8_t> seen((size_t)h.n_tris, 0);
    for (int32_t id : out.order) {
        if (id < 0 || id >= h.n_tris || seen[(size_t)id])
            return fail(RT_ERR_PARSE, "BVH cache holds an invalid triangle order");
        seen[(size_t)id] = 1;
    }
    // Topology is re-derived from the preorder leaf flags, as the JSON reader
    // does: depth and max_depth are recomputed (the stored ones are not
    // trusted), the flags must describe exactly one complete binary tree, every
    // node's range must lie inside the triangle array, and an interior node's
    // range must be the concatenation of its children's.
    out.max_depth = 0;
    std::vector<std::pair<int32_t, int32_t>> open_nodes;   // (interior node, children seen so far)
    for (int32_t i = 0; i < h.n_nodes; ++i) {
        if (i > 0 && open_nodes.empty()) return fail(RT_ERR_PARSE, "BVH cache: nodes after the end of the tree");
        const int32_t depth = (int32_t)open_nodes.size();
        const int32_t s = out.start[(size_t)i], e = out.end[(size_t)i];
        const bool empty_ok = h.n_tris == 0 && i == 0;
        if (!empty_ok && (s < 0 || e < s || e >= h.n_tris))
            return fail(RT_ERR_PARSE, "BVH cache: node " + std::to_string(i) + " has a triangle range outside the array");
        if (!open_nodes.empty()) {
            const int32_t parent = open_nodes.back().first;
            const bool first = open_nodes.back().second == 0;
            if (first ? s != out.start[(size_t)parent] : e != out.end[(size_t)parent])
                return fail(RT_ERR_PARSE, "BVH cache: node " + std::to_string(i) + " does not continue its parent's range");
            if (!first) {
                // the right child starts where the left one ended
                int32_t left = parent + 1;
                if (s != out.end[(size_t)left] + 1)
                    return fail(RT_ERR_PARSE, "BVH cache: children of node " + std::to_string(parent) + " do not tile its range");
            }
            open_nodes.back().second++;
        }
        out.depth[(size_t)i] = depth;
        out.max_depth = std::max(out.max_depth, depth);
        if (out.is_leaf[(size_t)i] != 0 && out.is_leaf[(size_t)i] != 1)
            return fail(RT_ERR_PARSE, "BVH cache: bad leaf flag");
        if (!out.is_leaf[(size_t)i]) {
            open_nodes.emplace_back(i, 0);
        } else {
            while (!open_nodes.empty() && open_nodes.back().second == 2) open_nodes.pop_back();
        }
    }
    if (!open_nodes.empty()) return fail(RT_ERR_PARSE, "BVH cache: the tree is truncated");
    return RT_OK;
}

}  // namespace rt
