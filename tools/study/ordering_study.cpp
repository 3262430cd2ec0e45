// ordering_study.cpp -- counts what variants of the ordered traversal's step
// would visit, on the CPU, before any of them is written as a kernel.
//
// The ordered kernel (rt_kernels.hip: k_trace_fast) sorts the four children of
// a record by entry distance and leaves up to three (child, distance) entries on
// the ray's stack.  The variants studied here give that up for less work per
// step, in the manner of compressed wide BVHs:
//   policy 0  what the kernel does today (rt_core.h: traverse_ordered)
//   policy 1  children visited in a fixed order per ray octant (front to back
//             along the octant's diagonal, by the centres of the child boxes),
//             and ONE stack entry per record: (record, mask of the children
//             still to visit, smallest entry distance among them); a popped
//             child is entered without knowing its own entry distance
//   policy 2  one entry per record as in 1, but children in distance order
//   policy 3  octant order as in 1, but one (child, distance) entry per child
//   policy 4  policy 0 restated in this file's loop (distance order, one entry per
//             child), which can also count the visits of every record: how many
//             of a ray's record fetches a table of the K most visited (or largest)
//             records would serve
// Output per policy: records visited and triangles fetched per ray, and the
// hits (which must not differ in distance from policy 0's).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../raytracer_amd/csrc/rt_core.h"

using namespace rtc;

namespace {

struct Step {
    bool hit[4];
    float tn[4];
    uint32_t ref[4];
    float centre[4][3];
};

void box_step(const DevScene &s, uint32_t rec, const Ray<float> &r, float h_t, Step &o) {
    const float *q = reinterpret_cast<const float *>(s.wide) + (size_t)kWideFloats * rec;
    const float inv[3] = {r.ix, r.iy, r.iz}, org[3] = {r.ox, r.oy, r.oz};
    uint32_t pn[3], pf[3];
    for (int a = 0; a < 3; ++a) {
        const uint32_t pl = (uint32_t)f2i(q[6 + a]), ph = (uint32_t)f2i(q[9 + a]);
        pn[a] = inv[a] < 0 ? ph : pl;
        pf[a] = inv[a] < 0 ? pl : ph;
    }
    for (int k = 0; k < 4; ++k) {
        float tna[3], tfa[3];
        for (int a = 0; a < 3; ++a) {
            const float bn = fmaf((float)((pn[a] >> (8 * k)) & 0xFFu), q[3 + a], q[a]);
            const float bf = fmaf((float)((pf[a] >> (8 * k)) & 0xFFu), q[3 + a], q[a]);
            tna[a] = (bn - org[a]) * inv[a];
            tfa[a] = (bf - org[a]) * inv[a];
            o.centre[k][a] = 0.5f * (bn + bf);
        }
        const float tn = fmaxf(fmaxf(tna[0], tna[1]), tna[2]);
        const float tf = fminf(fminf(tfa[0], tfa[1]), tfa[2]);
        o.ref[k] = (uint32_t)f2i(q[12 + k]);
        o.hit[k] = (tn <= fminf(tf, h_t * kTieSlack)) & (tf > 0);
        o.tn[k] = tn;
    }
}

struct Group {
    uint32_t ref[4];     // children still to visit, in visiting order
    float tn[4];
    int n, at;
    float tmin;          // smallest entry distance among ref[at..n)
};

void leaf(const DevScene &s, uint32_t cur, const Ray<float> &r, Hit<float> &h, uint64_t &tris) {
    const int32_t start = (int32_t)(cur & 0xFFFFFFu), count = (int32_t)((cur >> 24) & 0x7Fu);
    for (int32_t k = 0; k < count; ++k) {
        const f4 *q = s.tris + 3 * (size_t)(start + k);
        ++tris;
        tri_test<float, true>(r, q[0], q[1], q[2], start + k, h);
    }
}

// policies 1..3
void traverse_variant(const DevScene &s, const Ray<float> &r, Hit<float> &h, int policy, uint64_t &records,
                      uint64_t &tris, uint64_t &stale_children, uint32_t *visits) {
    std::vector<Group> stack;
    const bool octant = policy == 1 || policy == 3, per_child = policy == 3 || policy == 4;
    const float diag[3] = {r.ix < 0 ? -1.0f : 1.0f, r.iy < 0 ? -1.0f : 1.0f, r.iz < 0 ? -1.0f : 1.0f};
    uint32_t cur = s.fast_root;
    auto pop = [&]() -> uint32_t {
        while (!stack.empty()) {
            Group &g = stack.back();
            if (g.tmin > h.t * kTieSlack) {       // nothing in the group can still matter
                stack.pop_back();
                continue;
            }
            const int k = g.at++;
            const uint32_t ref = g.ref[k];
            const float tn = g.tn[k];
            if (g.at == g.n) stack.pop_back();
            if (per_child) {
                if (tn > h.t * kTieSlack) continue;           // per-child distances: stale children are dropped
            } else if (tn > h.t * kTieSlack) {
                ++stale_children;                              // entered although it cannot hold a closer hit
            }
            return ref;
        }
        return kRefDone;
    };
    for (;;) {
        while (cur != kRefDone && (cur & kLeafBit) == 0) {
            ++records;
            if (visits) ++visits[cur];
            Step st;
            box_step(s, cur, r, h.t, st);
            int idx[4] = {0, 1, 2, 3};
            if (octant) {
                float key[4];
                for (int k = 0; k < 4; ++k)
                    key[k] = st.centre[k][0] * diag[0] + st.centre[k][1] * diag[1] + st.centre[k][2] * diag[2];
                std::stable_sort(idx, idx + 4, [&](int a, int b) { return key[a] < key[b]; });
            } else {
                std::stable_sort(idx, idx + 4, [&](int a, int b) { return st.tn[a] < st.tn[b]; });
            }
            Group g{};
            g.n = 0;
            g.at = 0;
            for (int k = 0; k < 4; ++k)
                if (st.hit[idx[k]]) {
                    g.ref[g.n] = st.ref[idx[k]];
                    g.tn[g.n] = st.tn[idx[k]];
                    ++g.n;
                }
            if (g.n == 0) {
                cur = pop();
                continue;
            }
            cur = g.ref[0];
            if (g.n > 1) {
                g.at = 1;
                if (per_child) {
                    // one entry per child, farthest in visiting order first
                    for (int k = g.n - 1; k >= 1; --k) {
                        Group one{};
                        one.n = 1;
                        one.ref[0] = g.ref[k];
                        one.tn[0] = g.tn[k];
                        one.tmin = g.tn[k];
                        stack.push_back(one);
                    }
                } else {
                    g.tmin = g.tn[1];
                    for (int k = 2; k < g.n; ++k) g.tmin = fminf(g.tmin, g.tn[k]);
                    stack.push_back(g);
                }
            }
        }
        if (cur == kRefDone) return;
        leaf(s, cur, r, h, tris);
        cur = pop();
    }
}

}  // namespace

extern "C" void ordering_study(const float *wide, uint32_t fast_root, const float *tris, int32_t n_tris,
                               const float *rays8, int64_t n, int policy, float *hit_t, int32_t *hit_pos,
                               uint64_t *out, uint32_t *visits) {
    DevScene s{};
    s.wide = reinterpret_cast<const f4 *>(wide);
    s.fast_root = fast_root;
    s.tris = reinterpret_cast<const f4 *>(tris);
    s.n_tris = n_tris;
    uint64_t records = 0, tri_count = 0, stale = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float *q = rays8 + 8 * i;
        Ray<float> r;
        ray_init(r, q[0], q[1], q[2], q[4], q[5], q[6]);
        Hit<float> h{q[3], 0, 0, -1};
        if (policy == 0) {
            Counters c;
            traverse_ordered<true>(s, r, h, &c);
            records += c.nodes_visited;
            tri_count += c.tris_fetched;
        } else {
            traverse_variant(s, r, h, policy, records, tri_count, stale, visits);
        }
        hit_t[i] = h.t;
        hit_pos[i] = h.pos;
    }
    out[0] = records;
    out[1] = tri_count;
    out[2] = stale;
}
