// bvh_build.cpp -- SAH BVH construction with the reference's exact semantics
// (models/bvh.go:31-114 buildBVHNode, :146-212 GetSplitPlaneSAH;
// models/triangle.go:67-76 TriangleSorter, :140-168 GetTriangleBounds).
//
// The reference re-sorts a node's triangle range with sort.Slice once per axis
// and once more for the chosen plane, at every node: O(n log^2 n).  Its sort
// key (centre coordinate, then Triangle.Index) is a total order whenever the
// pair is unique, so the outcome of each of those sorts is independent of the
// sorting algorithm and of the prior arrangement.  This builder therefore
// sorts the triangles once per axis up front and keeps the three sorted views
// consistent by stable partitioning while descending -- O(n log n), ranges of
// different subtrees are disjoint so subtrees build on separate threads --
// and produces the same tree, bounds and triangle order.  Pairs that tie on
// (coordinate, Index) -- possible only across OBJ groups, where the reference
// itself leaves the order to an unstable sort -- are ordered by input
// position.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "gomath.h"
#include "rt_internal.h"

namespace rt {
namespace {

struct Pre {
    float c[3];    // Triangle.Center(): ((v0 + v1) + v2) * float32(1.0/3.0)
    float mn[3];   // the triangle's bounds, folded vertex by vertex with '<' / '>' as
    float mx[3];   // GetTriangleBounds folds them (the first of equal values is kept)
    int32_t index; // Triangle.Index
};

// One triangle as a sorted view holds it: everything a sweep or a partition of
// the view needs, so that both stream through memory instead of gathering
// 40-byte records by triangle id (at ten million triangles the gathers miss
// every cache and cost more than the arithmetic).
struct Item {
    float mn[3], mx[3];
    float c;       // centre coordinate on the view's own axis
    int32_t id;
};

inline bool item_less(const Pre *pre, const Item &a, const Item &b) {
    if (a.c == b.c) {
        const int32_t ia = pre[a.id].index, ib = pre[b.id].index;
        if (ia != ib) return ia < ib;
        return a.id < b.id;
    }
    return a.c < b.c;
}

struct BNode {
    int32_t start = 0, end = 0, depth = 0;
    float bmin[3], bmax[3];
    float split[4] = {0, 0, 0, 0};
    std::unique_ptr<BNode> left, right;
};

struct Builder {
    const rt_mesh &mesh;
    bool use_bvh;
    int32_t max_leaf, max_depth;
    int32_t n;
    std::vector<Pre> pre;
    std::vector<Item> view[3];      // triangles sorted by (c[a], index, id), node ranges contiguous
    std::vector<Item> tmp[3];       // partition scratch per axis, indexed by array position
    std::vector<int32_t> arr;       // the reference's triangle array (ids); a range is written when its leaf is made
    std::vector<uint8_t> side;      // per id: 1 = goes right
    std::vector<float> lsz[3], rsz[3];   // sweep areas per axis, indexed by array position
    std::atomic<int> spare_threads{0};   // threads subtrees may still take

    Builder(const rt_mesh &m, bool use, int32_t leaf, int32_t depth)
        : mesh(m), use_bvh(use), max_leaf(leaf), max_depth(depth), n(m.n_tris) {}

    static float area(const float mn[3], const float mx[3]) {
        float d0 = mx[0] - mn[0], d1 = mx[1] - mn[1], d2 = mx[2] - mn[2];
        return 2 * (d0 * d1 + d0 * d2 + d1 * d2);
    }

    void precompute() {
        pre.resize((size_t)n);
        const float third = (float)(1.0 / 3.0);
        for (int32_t i = 0; i < n; ++i) {
            const float *v = mesh.verts + 9 * (size_t)i;
            Pre &p = pre[(size_t)i];
            for (int k = 0; k < 3; ++k) {
                p.c[k] = ((v[k] + v[3 + k]) + v[6 + k]) * third;
                p.mn[k] = kF32Max;
                p.mx[k] = -kF32Max;
            }
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) {
                    const float c = v[3 * j + k];
                    if (c < p.mn[k]) p.mn[k] = c;
                    if (c > p.mx[k]) p.mx[k] = c;
                }
            p.index = mesh.group_index ? mesh.group_index[i] : i;
        }
    }

    void presort_axis(int a) {
        std::vector<Item> &items = view[a];
        items.resize((size_t)n);
        for (int32_t i = 0; i < n; ++i) {
            const Pre &p = pre[(size_t)i];
            Item &it = items[(size_t)i];
            std::memcpy(it.mn, p.mn, 12);
            std::memcpy(it.mx, p.mx, 12);
            it.c = p.c[a];
            it.id = i;
        }
        const Pre *pp = pre.data();
        // the sort itself on up to four threads: sorted quarters, merged pairwise
        const int parts = n > 400000 ? 4 : 1;
        if (parts == 1) {
            std::sort(items.begin(), items.end(), [pp](const Item &x, const Item &y) { return item_less(pp, x, y); });
            return;
        }
        size_t cut[5];
        for (int k = 0; k <= parts; ++k) cut[k] = (size_t)n * (size_t)k / (size_t)parts;
        std::vector<std::function<void()>> jobs;
        for (int k = 0; k < parts; ++k)
            jobs.push_back([&, k] {
                std::sort(items.begin() + (ptrdiff_t)cut[k], items.begin() + (ptrdiff_t)cut[k + 1],
                          [pp](const Item &x, const Item &y) { return item_less(pp, x, y); });
            });
        run_together(jobs);
        auto less = [pp](const Item &x, const Item &y) { return item_less(pp, x, y); };
        run_together({[&] { std::inplace_merge(items.begin(), items.begin() + (ptrdiff_t)cut[1], items.begin() + (ptrdiff_t)cut[2], less); },
                      [&] { std::inplace_merge(items.begin() + (ptrdiff_t)cut[2], items.begin() + (ptrdiff_t)cut[3], items.end(), less); }});
        std::inplace_merge(items.begin(), items.begin() + (ptrdiff_t)cut[2], items.end(), less);
    }

    // GetTriangleBounds over the array range.  The reference folds the vertices of
    // the range's triangles in array order with '<' / '>'; folding each triangle's
    // own (equally folded) bounds in the same order keeps the same elements.
    // order_axis >= 0: the range's array order is that view's; -1: input order
    // (the root); -2: the order written in arr[].
    void range_bounds(int32_t start, int32_t end, int order_axis, float mn[3], float mx[3]) const {
        for (int k = 0; k < 3; ++k) {
            mn[k] = kF32Max;
            mx[k] = -kF32Max;
        }
        auto fold = [&](const float *tmn, const float *tmx) {
            for (int k = 0; k < 3; ++k) {
                if (tmn[k] < mn[k]) mn[k] = tmn[k];
                if (tmx[k] > mx[k]) mx[k] = tmx[k];
            }
        };
        if (order_axis >= 0) {
            const Item *it = view[order_axis].data();
            for (int32_t i = start; i <= end; ++i) fold(it[i].mn, it[i].mx);
        } else {
            for (int32_t i = start; i <= end; ++i) {
                const Pre &p = pre[(size_t)(order_axis == -1 ? i : arr[(size_t)i])];
                fold(p.mn, p.mx);
            }
        }
    }

    // GetSplitPlaneSAH: returns the axis (0..2) or -1 when no candidate beat
    // MaxFloat32 (plane stays the zero vector).
    int sah_plane(int32_t start, int32_t end, float plane[4]) {
        // one sweep per axis; the reference keeps ONE running minimum over the
        // three axes in turn, so an axis only wins with a strictly lower cost
        // than the axes before it: the per-axis minima (first index reaching
        // it) compared in axis order give the same choice, and large nodes
        // sweep their axes at the same time
        float best_cost[3], best_d[3];
        auto sweep = [&](int a) {
            const Item *it = view[a].data();
            float *ls = lsz[a].data(), *rs = rsz[a].data();
            float mn[3], mx[3];
            std::memcpy(mn, it[start].mn, 12);
            std::memcpy(mx, it[start].mx, 12);
            for (int32_t i = start; i <= end; ++i) {
                const Item &p = it[i];
                for (int k = 0; k < 3; ++k) {
                    mn[k] = (float)gomath::go_min(mn[k], p.mn[k]);
                    mx[k] = (float)gomath::go_max(mx[k], p.mx[k]);
                }
                ls[(size_t)i] = area(mn, mx);
            }
            std::memcpy(mn, it[end].mn, 12);
            std::memcpy(mx, it[end].mx, 12);
            for (int32_t i = end; i >= start; --i) {
                const Item &p = it[i];
                for (int k = 0; k < 3; ++k) {
                    mn[k] = (float)gomath::go_min(mn[k], p.mn[k]);
                    mx[k] = (float)gomath::go_max(mx[k], p.mx[k]);
                }
                rs[(size_t)i] = area(mn, mx);
            }
            // cost as written in the reference: left area of [start..i] weighted
            // by (i-start), right area of [i+1..end] weighted by (end-i+1)
            float lowest = kF32Max, d = 0;
            for (int32_t i = start + 1; i < end; ++i) {
                float cost = ls[(size_t)i] * (float)(i - start) + rs[(size_t)i + 1] * (float)(end - i + 1);
                if (cost < lowest) {
                    lowest = cost;
                    d = (it[i].c + it[i + 1].c) / 2.0f;
                }
            }
            best_cost[a] = lowest;
            best_d[a] = d;
        };
        if (end - start + 1 > 65536)
            run_together({[&] { sweep(0); }, [&] { sweep(1); }, [&] { sweep(2); }});
        else
            for (int a = 0; a < 3; ++a) sweep(a);
        float lowest = kF32Max;
        int best_axis = -1;
        plane[0] = plane[1] = plane[2] = plane[3] = 0;
        for (int a = 0; a < 3; ++a)
            if (best_cost[a] < lowest) {
                lowest = best_cost[a];
                best_axis = a;
                plane[3] = best_d[a];
            }
        if (best_axis >= 0) plane[best_axis] = 1;
        return best_axis;
    }

    void partition_view(int a, int32_t start, int32_t end, int32_t split) {
        Item *it = view[a].data();
        Item *t = tmp[a].data();
        int32_t l = start, r = split;
        for (int32_t i = start; i <= end; ++i) {
            if (side[(size_t)it[i].id])
                t[(size_t)r++] = it[i];
            else
                t[(size_t)l++] = it[i];
        }
        std::memcpy(it + start, t + start, sizeof(Item) * (size_t)(end - start + 1));
    }

    std::unique_ptr<BNode> build(int32_t start, int32_t end, int32_t depth, int order_axis) {
        std::unique_ptr<BNode> node(new BNode);
        node->start = start;
        node->end = end;
        node->depth = depth;
        range_bounds(start, end, order_axis, node->bmin, node->bmax);
        int32_t count = end - start + 1;
        if (!(use_bvh && count > max_leaf && depth < max_depth)) {
            // a leaf keeps the array order its parent's split gave the range
            if (order_axis >= 0)
                for (int32_t i = start; i <= end; ++i) arr[(size_t)i] = view[order_axis][(size_t)i].id;
            return node;
        }

        float plane[4];
        int axis = sah_plane(start, end, plane);
        std::memcpy(node->split, plane, sizeof plane);
        // TriangleSorter(plane.xyz): the array range takes the order of that axis
        int child_order = axis;
        if (axis < 0) {
            // zero plane: every key is 0, order by (Index, input position)
            if (order_axis >= 0)
                for (int32_t i = start; i <= end; ++i) arr[(size_t)i] = view[order_axis][(size_t)i].id;
            else if (order_axis == -1)
                for (int32_t i = start; i <= end; ++i) arr[(size_t)i] = i;
            std::sort(arr.begin() + start, arr.begin() + end + 1, [&](int32_t x, int32_t y) {
                if (pre[(size_t)x].index != pre[(size_t)y].index) return pre[(size_t)x].index < pre[(size_t)y].index;
                return x < y;
            });
            child_order = -2;
        }
        auto id_at = [&](int32_t i) { return axis >= 0 ? view[axis][(size_t)i].id : arr[(size_t)i]; };
        auto plane_side = [&](int32_t id) {
            const float *c = pre[(size_t)id].c;
            return (plane[0] * c[0] + plane[1] * c[1] + plane[2] * c[2]) > plane[3];
        };
        int32_t split = start;
        if (axis >= 0) {
            // one-hot plane: the dot product is the centre coordinate on that axis (plus zeros)
            const Item *it = view[axis].data();
            const bool side0 = it[start].c > plane[3];
            for (int32_t i = start + 1; i < end; ++i)
                if ((it[i].c > plane[3]) != side0) {
                    split = i;
                    break;
                }
        } else {
            const bool side0 = plane_side(arr[(size_t)start]);
            for (int32_t i = start + 1; i < end; ++i)
                if (plane_side(arr[(size_t)i]) != side0) {
                    split = i;
                    break;
                }
        }
        if (split == start || split == end) split = start + (end - start) / 2;

        for (int32_t i = start; i < split; ++i) side[(size_t)id_at(i)] = 0;
        for (int32_t i = split; i <= end; ++i) side[(size_t)id_at(i)] = 1;
        if (count > 65536 && axis >= 0) {
            const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
            run_together({[&] { partition_view(a1, start, end, split); },
                          [&] { partition_view(a2, start, end, split); }});
        } else {
            for (int a = 0; a < 3; ++a)
                if (a != axis) partition_view(a, start, end, split);
        }

        // ranges of different subtrees are disjoint: the right subtree builds on a thread of
        // its own while threads are to be had and the range is worth one
        bool forked = false;
        if (count > 32768) {
            if (spare_threads.fetch_sub(1) > 0) {
                forked = true;
                std::unique_ptr<BNode> left;
                run_together({[&] { node->right = build(split, end, depth + 1, child_order); },
                              [&] { left = build(start, split - 1, depth + 1, child_order); }});
                node->left = std::move(left);
            }
            spare_threads.fetch_add(1);
        }
        if (!forked) {
            node->left = build(start, split - 1, depth + 1, child_order);
            node->right = build(split, end, depth + 1, child_order);
        }
        return node;
    }
};

void flatten_preorder(const BNode *n, Bvh &out) {
    // iterative preorder (left before right)
    std::vector<const BNode *> stack;
    stack.push_back(n);
    while (!stack.empty()) {
        const BNode *cur = stack.back();
        stack.pop_back();
        out.is_leaf.push_back(cur->left ? 0 : 1);
        out.start.push_back(cur->start);
        out.end.push_back(cur->end);
        out.depth.push_back(cur->depth);
        out.max_depth = std::max(out.max_depth, cur->depth);
        for (int k = 0; k < 3; ++k) {
            out.bmin.push_back(cur->bmin[k]);
            out.bmax.push_back(cur->bmax[k]);
        }
        for (int k = 0; k < 4; ++k) out.split.push_back(cur->split[k]);
        if (cur->left) {
            stack.push_back(cur->right.get());
            stack.push_back(cur->left.get());
        }
    }
    out.n_nodes = (int32_t)out.is_leaf.size();
}

}  // namespace

rt_status check_mesh(const rt_mesh *mesh) {
    if (!mesh) return fail(RT_ERR_INVALID, "mesh is NULL");
    if (mesh->n_tris < 0) return fail(RT_ERR_INVALID, "mesh.n_tris is negative");
    if (mesh->n_tris > 0 && !mesh->verts) return fail(RT_ERR_INVALID, "mesh.verts is NULL");
    if (mesh->n_materials < 1 || !mesh->kd || !mesh->is_light)
        return fail(RT_ERR_INVALID, "mesh needs at least one material (kd, is_light)");
    for (size_t i = 0; i < (size_t)mesh->n_tris * 9; ++i)
        if (!std::isfinite(mesh->verts[i]))
            return fail(RT_ERR_INVALID, "mesh.verts holds a non-finite coordinate (triangle " +
                                            std::to_string(i / 9) + ")");
    if (mesh->material)
        for (int32_t i = 0; i < mesh->n_tris; ++i)
            if (mesh->material[i] < 0 || mesh->material[i] >= mesh->n_materials)
                return fail(RT_ERR_INVALID, "mesh.material id out of range (triangle " +
                                                std::to_string(i) + ")");
    return RT_OK;
}

rt_status build_bvh(const rt_mesh &mesh, bool use_bvh, int32_t max_leaf, int32_t max_depth,
                    Bvh &out) {
    if (mesh.n_tris <= 0) return fail(RT_ERR_INVALID, "cannot build a BVH over zero triangles");
    const auto t_begin = std::chrono::steady_clock::now();
    Builder b(mesh, use_bvh, max_leaf, max_depth);
    const int32_t n = mesh.n_tris;
    b.precompute();
    b.arr.resize((size_t)n);
    for (int32_t i = 0; i < n; ++i) b.arr[(size_t)i] = i;
    bool will_split = use_bvh && n > max_leaf && 0 < max_depth;
    if (will_split) {
        if (n > 200000) {
            run_together({[&] { b.presort_axis(2); }, [&] { b.presort_axis(0); }, [&] { b.presort_axis(1); }});
        } else {
            for (int a = 0; a < 3; ++a) b.presort_axis(a);
        }
        b.side.resize((size_t)n);
        for (int a = 0; a < 3; ++a) {
            b.tmp[a].resize((size_t)n);
            b.lsz[a].resize((size_t)n + 1);
            b.rsz[a].resize((size_t)n + 1);
        }
        // worker budget: what the environment grants (OMP_NUM_THREADS is how job schedulers
        // say it), else the machine's cores, at most 16
        unsigned hw = std::thread::hardware_concurrency();
        if (const char *e = std::getenv("OMP_NUM_THREADS")) {
            const long v = std::strtol(e, nullptr, 10);
            if (v > 0) hw = (unsigned)v;
        }
        if (hw == 0) hw = 4;
        if (hw > 16) hw = 16;
        b.spare_threads = (int)hw - 1;
    }
    const bool timing = std::getenv("RT_BUILD_TIMING") != nullptr;
    const auto t_sorted = std::chrono::steady_clock::now();
    std::unique_ptr<BNode> root = b.build(0, n - 1, 0, -1);
    const auto t_built = std::chrono::steady_clock::now();
    out = Bvh();
    out.n_tris = n;
    flatten_preorder(root.get(), out);
    out.order = std::move(b.arr);
    if (timing) {
        const auto t_end = std::chrono::steady_clock::now();
        auto secs = [](auto a, auto z) { return std::chrono::duration<double>(z - a).count(); };
        std::fprintf(stderr, "[rt build] %d triangles: precompute + presort %.2f s, tree %.2f s, preorder copy %.2f s\n", n,
                     secs(t_begin, t_sorted), secs(t_sorted, t_built), secs(t_built, t_end));
    }
    return RT_OK;
}

}  // namespace rt
