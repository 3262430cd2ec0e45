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
#include <cmath>
#include <cstring>
#include <memory>

#include "gomath.h"
#include "rt_internal.h"

namespace rt {
namespace {

struct Pre {
    float c[3];    // Triangle.Center(): ((v0 + v1) + v2) * float32(1.0/3.0)
    float mn[3];   // Triangle.Min() via utility.Vec3Min
    float mx[3];   // Triangle.Max() via utility.Vec3Max
    int32_t index; // Triangle.Index
};

struct SortItem {
    float key;
    int32_t index;
    int32_t id;
};

inline bool item_less(const SortItem &a, const SortItem &b) {
    if (a.key == b.key) {
        if (a.index != b.index) return a.index < b.index;
        return a.id < b.id;
    }
    return a.key < b.key;
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
    std::vector<int32_t> view[3];   // ids sorted by (c[a], index, id), node ranges contiguous
    std::vector<int32_t> arr;       // the reference's triangle array (ids)
    std::vector<int32_t> tmp[3];    // partition scratch per axis, indexed by array position
    std::vector<uint8_t> side;      // per id: 1 = goes right
    std::vector<float> lsz[3], rsz[3];   // sweep areas per axis, indexed by array position

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
                p.mn[k] = (float)gomath::go_min(gomath::go_min(v[k], v[3 + k]), v[6 + k]);
                p.mx[k] = (float)gomath::go_max(gomath::go_max(v[k], v[3 + k]), v[6 + k]);
            }
            p.index = mesh.group_index ? mesh.group_index[i] : i;
        }
    }

    void presort_axis(int a) {
        std::vector<SortItem> items((size_t)n);
        for (int32_t i = 0; i < n; ++i) items[(size_t)i] = {pre[(size_t)i].c[a], pre[(size_t)i].index, i};
        std::sort(items.begin(), items.end(), item_less);
        view[a].resize((size_t)n);
        for (int32_t i = 0; i < n; ++i) view[a][(size_t)i] = items[(size_t)i].id;
    }

    // GetTriangleBounds over the array range, vertex by vertex with '<' / '>'
    void range_bounds(int32_t start, int32_t end, float mn[3], float mx[3]) const {
        for (int k = 0; k < 3; ++k) {
            mn[k] = kF32Max;
            mx[k] = -kF32Max;
        }
        for (int32_t i = start; i <= end; ++i) {
            const float *v = mesh.verts + 9 * (size_t)arr[(size_t)i];
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) {
                    float c = v[3 * j + k];
                    if (c < mn[k]) mn[k] = c;
                    if (c > mx[k]) mx[k] = c;
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
            const int32_t *ids = view[a].data();
            float *ls = lsz[a].data(), *rs = rsz[a].data();
            float mn[3], mx[3];
            std::memcpy(mn, pre[(size_t)ids[start]].mn, 12);
            std::memcpy(mx, pre[(size_t)ids[start]].mx, 12);
            for (int32_t i = start; i <= end; ++i) {
                const Pre &p = pre[(size_t)ids[i]];
                for (int k = 0; k < 3; ++k) {
                    mn[k] = (float)gomath::go_min(mn[k], p.mn[k]);
                    mx[k] = (float)gomath::go_max(mx[k], p.mx[k]);
                }
                ls[(size_t)i] = area(mn, mx);
            }
            std::memcpy(mn, pre[(size_t)ids[end]].mn, 12);
            std::memcpy(mx, pre[(size_t)ids[end]].mx, 12);
            for (int32_t i = end; i >= start; --i) {
                const Pre &p = pre[(size_t)ids[i]];
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
                    d = (pre[(size_t)ids[i]].c[a] + pre[(size_t)ids[i + 1]].c[a]) / 2.0f;
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
        int32_t *ids = view[a].data();
        int32_t *t = tmp[a].data();
        int32_t l = start, r = split;
        for (int32_t i = start; i <= end; ++i) {
            int32_t id = ids[i];
            if (side[(size_t)id])
                t[(size_t)r++] = id;
            else
                t[(size_t)l++] = id;
        }
        std::memcpy(ids + start, t + start, sizeof(int32_t) * (size_t)(end - start + 1));
    }

    std::unique_ptr<BNode> build(int32_t start, int32_t end, int32_t depth) {
        std::unique_ptr<BNode> node(new BNode);
        node->start = start;
        node->end = end;
        node->depth = depth;
        range_bounds(start, end, node->bmin, node->bmax);
        int32_t count = end - start + 1;
        if (!(use_bvh && count > max_leaf && depth < max_depth)) return node;

        float plane[4];
        int axis = sah_plane(start, end, plane);
        std::memcpy(node->split, plane, sizeof plane);
        // TriangleSorter(plane.xyz): the array range takes the order of that axis
        if (axis >= 0) {
            std::memcpy(arr.data() + start, view[axis].data() + start, sizeof(int32_t) * (size_t)count);
        } else {
            // zero plane: every key is 0, order by (Index, input position)
            std::sort(arr.begin() + start, arr.begin() + end + 1, [&](int32_t x, int32_t y) {
                if (pre[(size_t)x].index != pre[(size_t)y].index) return pre[(size_t)x].index < pre[(size_t)y].index;
                return x < y;
            });
        }
        auto plane_side = [&](int32_t id) {
            const float *c = pre[(size_t)id].c;
            return (plane[0] * c[0] + plane[1] * c[1] + plane[2] * c[2]) > plane[3];
        };
        int32_t split = start;
        bool side0 = plane_side(arr[(size_t)start]);
        for (int32_t i = start + 1; i < end; ++i) {
            if (plane_side(arr[(size_t)i]) != side0) {
                split = i;
                break;
            }
        }
        if (split == start || split == end) split = start + (end - start) / 2;

        for (int32_t i = start; i < split; ++i) side[(size_t)arr[(size_t)i]] = 0;
        for (int32_t i = split; i <= end; ++i) side[(size_t)arr[(size_t)i]] = 1;
        if (count > 65536 && axis >= 0) {
            const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
            run_together({[&] { partition_view(a1, start, end, split); },
                          [&] { partition_view(a2, start, end, split); }});
        } else {
            for (int a = 0; a < 3; ++a)
                if (a != axis) partition_view(a, start, end, split);
        }

        if (depth < 4 && count > 65536) {
            std::unique_ptr<BNode> left;
            run_together({[&] { node->right = build(split, end, depth + 1); },
                          [&] { left = build(start, split - 1, depth + 1); }});
            node->left = std::move(left);
        } else {
            node->left = build(start, split - 1, depth + 1);
            node->right = build(split, end, depth + 1);
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
    }
    std::unique_ptr<BNode> root = b.build(0, n - 1, 0);
    out = Bvh();
    out.n_tris = n;
    flatten_preorder(root.get(), out);
    out.order = std::move(b.arr);
    return RT_OK;
}

}  // namespace rt
