// rt_core.h -- the per-ray and per-path-vertex arithmetic of the kernels,
// written as host/device inline functions so that the HIP kernels
// (rt_kernels.hip) and a lane-serial CPU harness used by the unit tests
// compile the very same code.
//
// Reference semantics implemented here (paths under src/backend/):
//   models/ray.go:23-42        ray set-up (invDir, sign)
//   models/aabb.go:62-99       slab test, compare-and-assign form
//   models/triangle.go:78-103  one-sided Moller-Trumbore, det < 1e-4
//   models/bvh.go:117-144      WalkNode order: left subtree completely before
//                              the right child's box is tested, strict '<'
//   process/trace.go:44-109    per-vertex shading / bounce arithmetic
//   utility/utility.go:34-70   ClampColor, RandomInHemisphere
//   models/arealight.go:42-56  AreaLight.Sample
//   models/camera.go:99-129    GetCameraRay
//   main.go:158-182            resolve
// Build with -ffp-contract=off: the reference's float32 operations are never
// fused, and the bit-exact modes depend on it.
#pragma once

#include <cstdint>

#include "gomath.h"

namespace rtc {

constexpr float kF32Max = 3.40282346638528859811704183484516925440e+38f;
constexpr float kF32Inf = __builtin_huge_valf();

struct alignas(16) f4 {
    float x, y, z, w;
};

RT_HD int f2i(float f) {
    union { float f; int i; } c;
    c.f = f;
    return c.i;
}
RT_HD float i2f(int i) {
    union { float f; int i; } c;
    c.i = i;
    return c.f;
}

// ---------------------------------------------------------------------------
// Flattened scene, resident in HBM.
//
// nodes: 2 x f4 per node, preorder (BVHNode.Index order).  The reference walks
// left-child-first and tests a child's box from its parent; in preorder that is
// "test node i; on pass continue at i+1, on fail jump past i's subtree".  Each
// node therefore stores its own box plus that jump target ("skip"):
//   [0] = (min.x, min.y, min.z, bits(a))   [1] = (max.x, max.y, max.z, bits(b))
//   interior: a = skip index, b = -1       leaf: a = first triangle, b = count
// tris: 3 x f4 per triangle in sorted-array order:
//   [0] = (N.x, N.y, N.z, V0.x) [1] = (V0.y, V0.z, E0.x, E0.y) [2] = (E0.z, W.x, W.y, W.z)
//   with W = v0v2 = Edge2 * -1.
// ---------------------------------------------------------------------------
struct DevScene {
    const f4 *nodes;
    // The same reference-order tree with explicit links, free of the preorder
    // layout (may be null): (min.xyz, bits next)(max.xyz, bits w).  Interior:
    // next = where to go when the box fails, w = -1 - index of the left child.
    // Leaf: next = where to go after the leaf, w = count << 24 | first triangle.
    // The n_top nodes with the largest boxes (a top of the tree) come first in
    // the array so that a block can keep them in LDS.
    const f4 *linked;
    int32_t n_top;
    const f4 *tris;
    // What shading needs of a triangle, one 16-byte word per array position so
    // that a path vertex costs one scattered fetch: (N.x, N.y, N.z, bits(m)),
    // m = material id, bit 30 set when the material is "Light".
    const f4 *tri_shade;
    const float *kd;            // n_materials * 3
    const float *tri_uv;        // n_tris * 6 or nullptr
    const int32_t *mat_tex;     // n_materials texture slot or -1 (nullptr: none)
    const uint8_t *const *tex_data;   // per slot RGBA8 rows
    const int32_t *tex_dim;     // per slot (w, h)
    // Ordered-traversal view of the same tree (fast mode): 4-wide records of 64
    // bytes (16 dwords):
    //   [0..2]  origin of the record's frame (lower corner of the children's union)
    //   [3..5]  step per axis (extent / 255, nudged up until every plane decodes outward)
    //   [6..8]  lower planes x, y, z: child c's 8-bit offset in byte c
    //   [9..11] upper planes x, y, z, likewise
    //   [12..15] child refs: interior -> record index; leaf -> kLeafBit | count << 24
    //            | first triangle; unused slot -> kLeafBit (count 0)
    // The kWideTopRecords records with the largest frames (a top of the tree) come
    // first, largest first; the rest follow in depth-first order.
    // A plane decodes as fma(offset, step, origin); decoded boxes contain the
    // binary tree's boxes (lower planes rounded down, upper planes up) with a
    // small margin to spare (flatten.cpp: margin) for the near-origin form.
    const f4 *wide;
    uint32_t fast_root;         // ref of the root (a record index, or a leaf ref)
    int32_t fast_ok;            // 0: tree does not fit the fast encoding
    int32_t n_wide;             // records in `wide`
    const double *cam_tab;      // n_cam * 2
    const double *light_tab;    // n_light * 2
    int32_t n_nodes, n_tris, n_cam, n_light;
};

constexpr int32_t kLightBit = 1 << 30;
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kRefDone = 0xFFFFFFFFu;   // empty-stack marker (count 127, start 2^24-1 is never built)
constexpr int32_t kFastSplitLeaf = 2;   // ordered view: leaf ranges up to this size are never cut (one leaf step tests two triangles)
constexpr int32_t kFastMaxLeaf = 4;     // ordered view: larger leaf ranges are always cut; between the two the builder weighs the cut
constexpr int32_t kWideFloats = 16;     // one ordered-traversal record: frame + 4 quantised child boxes + 4 refs = 64 bytes
constexpr int32_t kWideTopRecords = 1024;   // ordered view: this many records with the largest frames lead the array (flatten.cpp: top_records_first)
constexpr int32_t kFastStackEntries = 96;   // per-ray stack of the ordered traversal
constexpr int32_t kFastMaxTris = (1 << 24) - 2;
// Ordered traversal: a child box or stack entry is still entered when its entry
// distance lies up to this factor beyond the current hit distance.  Two
// triangles that share an edge can be hit at bit-equal distances, and the tie
// goes to the earlier array position (the one the reference's left-first order
// meets first, bvh.go:125), so the box of that earlier triangle has to be
// entered although a hit AT its distance is already known.  The box's entry
// distance and the triangle's hit distance come out of different float32
// computations and can disagree by a few rounding steps; 2^-18 (32 steps)
// covers that with a wide margin and admits no measurable number of extra
// boxes.  Triangles in an admitted box are still accepted only below the hit
// distance, or at it from an earlier position.  FLT_MAX times the factor is
// +inf, which admits every finite entry distance, as FLT_MAX itself does.
constexpr float kTieSlack = 1.0f + 0x1p-18f;
// Ordered traversal, near-origin form (RT_MODE_FAST_NEAR_F32): for rays that
// start within this many scene diagonals of the scene's bounding box the step
// subtracts the ray origin from the record's frame origin once per record and
// decodes every plane relative to the ray, (offset * step + (origin - o)) *
// invDir: two packed instructions per plane pair instead of three.  The
// flattener keeps every decoded plane far enough outside its box (flatten.cpp:
// margin) that the form stays on the safe side of the reference's box test for
// such rays.  Every ray of a path traced from a camera in that region
// qualifies: the others start on the scene's surfaces.
constexpr int kNearOriginDiagonals = 1;

// ---------------------------------------------------------------------------
// Ray state for the reference-order traversal.
// ---------------------------------------------------------------------------
template <typename Real>
struct Ray {
    Real ox, oy, oz, dx, dy, dz, ix, iy, iz;
    bool sx, sy, sz;
};

// The inverse direction of a ray (ray.go:33-37), three correctly rounded
// quotients 1 / d.  On the device a float32 component takes a short path: the
// hardware reciprocal (1 ulp) and two Newton steps in fused arithmetic, five
// instructions instead of the ten of the compiler's division, which also scales
// denormal and huge operands.  The short form equals the IEEE quotient for EVERY
// float whose magnitude lies in [2^-100, 2^100], either sign (compared
// exhaustively on the GPU by tools/ubench/rcp_check.hip, which the GPU tests
// run); a wave in which any lane holds a component outside that range -- zero,
// denormal, infinite, NaN -- takes the division for all three, behind one vote.
template <typename Real>
RT_HD void recip3(Real x, Real y, Real z, Real &ix, Real &iy, Real &iz) {
    ix = (Real)1.0 / x;
    iy = (Real)1.0 / y;
    iz = (Real)1.0 / z;
}
#if defined(__HIP_DEVICE_COMPILE__)
RT_HD float recip_short(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
template <>
RT_HD void recip3<float>(float x, float y, float z, float &ix, float &iy, float &iz) {
    ix = recip_short(x);
    iy = recip_short(y);
    iz = recip_short(z);
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y), az = __builtin_fabsf(z);
    const bool plain = (ax >= 0x1p-100f) & (ax <= 0x1p100f) & (ay >= 0x1p-100f) & (ay <= 0x1p100f) &
                       (az >= 0x1p-100f) & (az <= 0x1p100f);
    if (__builtin_amdgcn_ballot_w64(!plain) != 0) {
        ix = 1.0f / x;
        iy = 1.0f / y;
        iz = 1.0f / z;
    }
}
#endif

template <typename Real>
RT_HD void ray_init(Ray<Real> &r, float ox, float oy, float oz, float dx, float dy, float dz) {
    r.ox = (Real)ox; r.oy = (Real)oy; r.oz = (Real)oz;
    r.dx = (Real)dx; r.dy = (Real)dy; r.dz = (Real)dz;
    recip3<Real>(r.dx, r.dy, r.dz, r.ix, r.iy, r.iz);
    r.sx = r.ix < 0; r.sy = r.iy < 0; r.sz = r.iz < 0;
}

// aabb.go:62-99.  Returns hit and writes the interval.
template <typename Real>
RT_HD bool slab(const Ray<Real> &r, const f4 &lo, const f4 &hi, Real &otmin, Real &otmax) {
    Real tmin = ((Real)(r.sx ? hi.x : lo.x) - r.ox) * r.ix;
    Real tmax = ((Real)(r.sx ? lo.x : hi.x) - r.ox) * r.ix;
    Real tymin = ((Real)(r.sy ? hi.y : lo.y) - r.oy) * r.iy;
    Real tymax = ((Real)(r.sy ? lo.y : hi.y) - r.oy) * r.iy;
    if (tmin > tymax || tymin > tmax) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    Real tzmin = ((Real)(r.sz ? hi.z : lo.z) - r.oz) * r.iz;
    Real tzmax = ((Real)(r.sz ? lo.z : hi.z) - r.oz) * r.iz;
    if (tmin > tzmax || tzmin > tmax) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    otmin = tmin;
    otmax = tmax;
    return true;
}

// 1 / det of the triangle test (triangle.go:86), correctly rounded.  On the
// device a float32 determinant takes a short path: the hardware reciprocal
// (1 ulp) and two Newton steps in fused arithmetic, five instructions instead
// of the ten of the compiler's division, which also scales denormal and huge
// operands.  The short form equals the IEEE quotient for EVERY float in
// [2^-100, 2^100] (compared exhaustively on the GPU by tools/ubench/rcp_check.hip,
// which the GPU tests run); the cull has rejected determinants below 1e-4
// before the quotient is used, and larger ones (or NaN) take the division.
template <typename Real>
RT_HD Real recip_det(Real det) {
    return (Real)1.0 / det;
}
#if defined(__HIP_DEVICE_COMPILE__)
template <>
RT_HD float recip_det<float>(float det) {
    float r = recip_short(det);
    // a vote of the wave, so that the division's ten instructions sit behind a scalar
    // branch that is all but never taken instead of being computed and masked out
    const bool wide_range = !(det <= 0x1p100f);
    if (__builtin_amdgcn_ballot_w64(wide_range) != 0)
        r = wide_range ? 1.0f / det : r;
    return r;
}
#endif

template <typename Real>
struct Hit {
    Real t, u, v;
    int32_t pos;
};

// bvh.go:120-131 for one triangle record: back-face cull, triangle.go:78-103,
// then "t > 0 && t < tmin" (first visited wins ties).  Returns true when the
// Moller-Trumbore test was evaluated (for the counters).
// TIE_BY_POS (fast mode only): an exactly equal t also wins when the triangle
// sits earlier in the array, which is what the reference's left-first order
// yields for ties, so the result does not depend on the visiting order.
template <typename Real, bool TIE_BY_POS = false>
RT_HD bool tri_test(const Ray<Real> &r, const f4 &q0, const f4 &q1, const f4 &q2, int32_t pos,
                    Hit<Real> &h) {
    const Real nx = (Real)q0.x, ny = (Real)q0.y, nz = (Real)q0.z;
    if (nx * r.dx + ny * r.dy + nz * r.dz > 0) return false;
    const Real wx = (Real)q2.y, wy = (Real)q2.z, wz = (Real)q2.w;
    const Real ex = (Real)q1.z, ey = (Real)q1.w, ez = (Real)q2.x;
    // pvec = D x v0v2
    const Real px = r.dy * wz - r.dz * wy;
    const Real py = r.dz * wx - r.dx * wz;
    const Real pz = r.dx * wy - r.dy * wx;
    const Real det = ex * px + ey * py + ez * pz;
    if (det < (Real)0.0001f) return true;
    const Real invdet = recip_det(det);
    const Real tx = r.ox - (Real)q0.w, ty = r.oy - (Real)q1.x, tz = r.oz - (Real)q1.y;
    const Real u = (tx * px + ty * py + tz * pz) * invdet;
    if (u < 0 || u > 1) return true;
    // qvec = tvec x Edge0
    const Real qx = ty * ez - tz * ey;
    const Real qy = tz * ex - tx * ez;
    const Real qz = tx * ey - ty * ex;
    const Real v = (r.dx * qx + r.dy * qy + r.dz * qz) * invdet;
    if (v < 0 || u + v > 1) return true;
    const Real t = (wx * qx + wy * qy + wz * qz) * invdet;
    if (t > 0 && (t < h.t || (TIE_BY_POS && t == h.t && pos < h.pos))) {
        h.t = t;
        h.u = u;
        h.v = v;
        h.pos = pos;
    }
    return true;
}

// The same test with no early exit: every value is computed and the conditions
// are combined at the end, each in the form the step-by-step version tests it
// (a NaN fails a comparison in both), so `inside` holds exactly when tri_test
// would reach its last line with t > 0, and t, u, v are then the same bits.
// For a wave whose lanes test different triangles the early exits skip nothing
// (some lane always goes on) and only cost the bookkeeping of nested branches.
// Returns whether the triangle faces the ray (the counters' "tested").
template <typename Real>
RT_HD bool tri_eval(const Ray<Real> &r, const f4 &q0, const f4 &q1, const f4 &q2, Real &t, Real &u,
                    Real &v, bool &inside) {
    const Real nx = (Real)q0.x, ny = (Real)q0.y, nz = (Real)q0.z;
    const bool facing = !(nx * r.dx + ny * r.dy + nz * r.dz > 0);
    const Real wx = (Real)q2.y, wy = (Real)q2.z, wz = (Real)q2.w;
    const Real ex = (Real)q1.z, ey = (Real)q1.w, ez = (Real)q2.x;
    const Real px = r.dy * wz - r.dz * wy;
    const Real py = r.dz * wx - r.dx * wz;
    const Real pz = r.dx * wy - r.dy * wx;
    const Real det = ex * px + ey * py + ez * pz;
    const Real invdet = recip_det(det);
    const Real tx = r.ox - (Real)q0.w, ty = r.oy - (Real)q1.x, tz = r.oz - (Real)q1.y;
    u = (tx * px + ty * py + tz * pz) * invdet;
    const Real qx = ty * ez - tz * ey;
    const Real qy = tz * ex - tx * ez;
    const Real qz = tx * ey - ty * ex;
    v = (r.dx * qx + r.dy * qy + r.dz * qz) * invdet;
    t = (wx * qx + wy * qy + wz * qz) * invdet;
    inside = facing & !(det < (Real)0.0001f) & !((u < 0) | (u > 1)) & !((v < 0) | (u + v > 1)) & (t > 0);
    return facing;
}

struct Counters {
    uint32_t nodes_visited = 0, box_tests = 0, tris_fetched = 0, tris_tested = 0;
};

// One whole reference-order traversal (lane-serial form; the kernels inline the
// same steps around their wave-level scheduling).  COUNT adds the counters.
template <typename Real, bool COUNT>
RT_HD void traverse_ref_order(const DevScene &s, const Ray<Real> &r, Hit<Real> &h, Counters *c) {
    int32_t node = 0;
    while (node < s.n_nodes) {
        const f4 lo = s.nodes[2 * node], hi = s.nodes[2 * node + 1];
        const int32_t a = f2i(lo.w), b = f2i(hi.w);
        bool pass = true;
        if (node != 0) {   // the root is entered without a box test (trace.go:139)
            Real t0, t1;
            if (COUNT) c->box_tests++;
            pass = slab(r, lo, hi, t0, t1) && t0 < h.t && t1 > 0;
        }
        if (b >= 0) {
            if (pass) {
                for (int32_t k = 0; k < b; ++k) {
                    const f4 *q = s.tris + 3 * (size_t)(a + k);
                    if (COUNT) c->tris_fetched++;
                    bool tested = tri_test(r, q[0], q[1], q[2], a + k, h);
                    if (COUNT && tested) c->tris_tested++;
                }
            }
            node = node + 1;
        } else {
            if (COUNT && pass) c->nodes_visited++;
            node = pass ? node + 1 : a;
        }
    }
}

// The fast mode's ordered traversal, lane-serial form: nearer child first, the
// farther one stacked with its entry distance and dropped at pop time once a
// closer hit exists (an entry at the hit distance, or within kTieSlack of it, is
// kept: it may hold a triangle that ties with the hit and sits earlier in the
// array).  Box slabs use the reference's (bound - origin) * invDir;
// the triangle test is tri_test() unchanged, so t, u, v of a given triangle are
// bit-identical to the reference order's.
// Sorts four (entry distance, child ref) pairs by distance, ascending, with a
// fixed five-exchange network; an exchange happens only on a strictly smaller
// distance, so of two equal children the earlier slot stays first where the
// network compares them directly.  Missed children carry +inf and end up last.
RT_HD void wide_sort4(float key[4], uint32_t ref[4]) {
#define RT_CE(a, b)                                    \
    do {                                               \
        const bool sw = key[b] < key[a];               \
        const float ka = sw ? key[b] : key[a];         \
        const float kb = sw ? key[a] : key[b];         \
        const uint32_t ra = sw ? ref[b] : ref[a];      \
        const uint32_t rb = sw ? ref[a] : ref[b];      \
        key[a] = ka; key[b] = kb; ref[a] = ra; ref[b] = rb; \
    } while (0)
    RT_CE(0, 1);
    RT_CE(2, 3);
    RT_CE(0, 2);
    RT_CE(1, 3);
    RT_CE(1, 2);
#undef RT_CE
}

template <bool COUNT, bool NEAR = false>
RT_HD void traverse_ordered(const DevScene &s, const Ray<float> &r, Hit<float> &h, Counters *c) {
    uint32_t stack_ref[kFastStackEntries];
    float stack_t[kFastStackEntries];
    int32_t sp = 0;
    uint32_t cur = s.fast_root;
    for (;;) {
        while ((cur & kLeafBit) == 0) {
            const float *q = reinterpret_cast<const float *>(s.wide) + (size_t)kWideFloats * (size_t)cur;
            if (COUNT) { c->nodes_visited++; c->box_tests += 4; }
            float key[4];
            uint32_t ref[4];
            int n_hit = 0;
            // near / far plane words per axis, chosen by the sign of the inverse direction
            const float inv[3] = {r.ix, r.iy, r.iz}, org[3] = {r.ox, r.oy, r.oz};
            uint32_t pn[3], pf[3];
            for (int a = 0; a < 3; ++a) {
                const uint32_t pl = (uint32_t)f2i(q[6 + a]), ph = (uint32_t)f2i(q[9 + a]);
                pn[a] = inv[a] < 0 ? ph : pl;
                pf[a] = inv[a] < 0 ? pl : ph;
            }
            // near-origin form: the frame origin relative to the ray, rounded once per record
            const float rel[3] = {q[0] - org[0], q[1] - org[1], q[2] - org[2]};
            for (int k = 0; k < 4; ++k) {
                float tna[3], tfa[3];
                for (int a = 0; a < 3; ++a) {
                    const float on = (float)((pn[a] >> (8 * k)) & 0xFFu), of = (float)((pf[a] >> (8 * k)) & 0xFFu);
                    if (NEAR) {
                        tna[a] = fmaf(on, q[3 + a], rel[a]) * inv[a];
                        tfa[a] = fmaf(of, q[3 + a], rel[a]) * inv[a];
                    } else {
                        tna[a] = (fmaf(on, q[3 + a], q[a]) - org[a]) * inv[a];
                        tfa[a] = (fmaf(of, q[3 + a], q[a]) - org[a]) * inv[a];
                    }
                }
                const float tn = fmaxf(fmaxf(tna[0], tna[1]), tna[2]);
                const float tf = fminf(fminf(tfa[0], tfa[1]), tfa[2]);
                ref[k] = (uint32_t)f2i(q[12 + k]);
                // an unused slot has an inverted box (tn > tf) and needs no test of its own
                const bool hit = (tn <= tf) & (tf > 0);
                key[k] = hit ? tn : kF32Inf;
            }
            wide_sort4(key, ref);
            // whether a child lies within reach of the hit distance is asked after the sort, as in
            // the kernel: missed children carry +inf, the reach is kept finite, one compare says both
            const float reach = fminf(h.t * kTieSlack, kF32Max);
            for (int k = 0; k < 4; ++k) n_hit += key[k] <= reach ? 1 : 0;
            if (n_hit == 0) {
                cur = kRefDone;
                while (sp > 0) {
                    sp--;
                    if (stack_t[sp] <= h.t * kTieSlack) { cur = stack_ref[sp]; break; }
                }
            } else {
                // farthest first, so that the nearest of the rest is popped first
                for (int k = n_hit - 1; k >= 1; --k) {
                    stack_ref[sp] = ref[k];
                    stack_t[sp] = key[k];
                    sp++;
                }
                cur = ref[0];
            }
        }
        if (cur == kRefDone) return;
        const int32_t start = (int32_t)(cur & 0xFFFFFFu);
        const int32_t count = (int32_t)((cur >> 24) & 0x7Fu);
        for (int32_t k = 0; k < count; ++k) {
            const f4 *q = s.tris + 3 * (size_t)(start + k);
            if (COUNT) c->tris_fetched++;
            const bool tested = tri_test<float, true>(r, q[0], q[1], q[2], start + k, h);
            if (COUNT && tested) c->tris_tested++;
        }
        cur = kRefDone;
        while (sp > 0) {
            sp--;
            if (stack_t[sp] <= h.t * kTieSlack) { cur = stack_ref[sp]; break; }
        }
        if (cur == kRefDone) return;
    }
}

// ---------------------------------------------------------------------------
// Counter RNG: one splitmix64 stream per camera sample; draw k of a stream is
// mix(stream + (k+1)*golden).  24-bit floats in [0,1) like Go's rand.Float32.
// ---------------------------------------------------------------------------
RT_HD uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ULL;
RT_HD uint64_t rng_stream(int64_t seed, uint64_t path_index) {
    uint64_t a = mix64((uint64_t)seed + kGolden);
    return mix64(a ^ (path_index * 0xD1B54A32D192ED03ULL + 0x8CB92BA72F3D8DD7ULL));
}
RT_HD uint64_t rng_u64(uint64_t stream, uint32_t counter) {
    return mix64(stream + ((uint64_t)counter + 1) * kGolden);
}
RT_HD float rng_f32(uint64_t stream, uint32_t counter) {
    return (float)(rng_u64(stream, counter) >> 40) * 0x1p-24f;
}
RT_HD uint64_t light_base(uint64_t stream) { return rng_u64(stream ^ 0xA5A5A5A5A5A5A5A5ULL, 0); }

// ---------------------------------------------------------------------------
// mgl32 float32 vector helpers
// ---------------------------------------------------------------------------
struct v3 {
    float x, y, z;
};
RT_HD float dot(const v3 &a, const v3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD float lensqr(const v3 &a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RT_HD float len(const v3 &a) { return (float)sqrt((double)lensqr(a)); }
RT_HD v3 normalize(const v3 &a) {
    float l = 1.0f / len(a);
    return v3{a.x * l, a.y * l, a.z * l};
}
RT_HD v3 transform_coordinate(const v3 &v, const float *m) {
    float t[4];
    for (int r = 0; r < 4; ++r) t[r] = m[r] * v.x + m[4 + r] * v.y + m[8 + r] * v.z + m[12 + r] * 1.0f;
    float iw = 1 / t[3];
    return v3{t[0] * iw, t[1] * iw, t[2] * iw};
}

// Everything a pass needs on the device, prepared once on the host
// (context.go:245-275 RenderPass.Initialize, camera.go:43-85).
struct PassParams {
    float cam_transform[16];
    float top_left[3];
    float steps[2];                 // horizontal, vertical
    float ppd;
    int32_t projection;
    int32_t spp;                    // Camera.RaysPerPixel
    int32_t x_offset, y_offset, width, height;
    int64_t rng_seed;
    int32_t L;                      // LightSampleRays
    int32_t B;                      // BounceLimit (uint8 range)
    float light_transform[16];
    float light_size[2];
    float light_emission[3];
    float light_normal[3];
    int32_t incremental;            // path index = s*pixels + i instead of i*spp + s
    // Several tiles of one frame traced together (same camera and settings):
    // n_tiles > 0 replaces x_offset/y_offset/width/height/rng_seed by a table.
    int32_t n_tiles;
    const struct TileDesc *tiles;
};

// One tile of a multi-tile pass; pixels of all tiles are numbered consecutively,
// tile after tile (pix_start = number of pixels in the tiles before this one).
struct TileDesc {
    int32_t x_offset, y_offset, width, height;
    int64_t rng_seed;
    uint32_t pix_start;
    uint32_t pad;
};

// Where a camera sample lives: its tile's offsets and seed, its pixel inside the
// tile and its path index (the reference's running ray index `ri`, main.go:132).
struct PathCtx {
    int32_t x_offset, y_offset, x, y;
    int64_t rng_seed;
    uint64_t path_index;
};

RT_HD PathCtx locate_path(const PassParams &p, uint32_t pix, uint32_t sample) {
    PathCtx c;
    int32_t w = p.width, h = p.height;
    uint32_t local = pix;
    c.x_offset = p.x_offset;
    c.y_offset = p.y_offset;
    c.rng_seed = p.rng_seed;
    if (p.n_tiles > 0) {
        int32_t lo = 0, hi = p.n_tiles - 1;   // last tile whose pix_start <= pix
        while (lo < hi) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (p.tiles[mid].pix_start <= pix) lo = mid; else hi = mid - 1;
        }
        const TileDesc t = p.tiles[lo];
        w = t.width;
        h = t.height;
        local = pix - t.pix_start;
        c.x_offset = t.x_offset;
        c.y_offset = t.y_offset;
        c.rng_seed = t.rng_seed;
    }
    c.x = (int32_t)(local % (uint32_t)w);
    c.y = (int32_t)(local / (uint32_t)w);
    const uint64_t pixels = (uint64_t)w * (uint64_t)h;
    c.path_index = p.incremental ? (uint64_t)sample * pixels + local
                                 : (uint64_t)local * (uint64_t)p.spp + sample;
    return c;
}

// camera.go:99-129
RT_HD void camera_ray(const DevScene &s, const PassParams &p, const PathCtx &c, v3 &origin,
                      v3 &dir) {
    const int32_t x = c.x, y = c.y;
    const uint64_t path_index = c.path_index;
    float jx = 0.5f, jy = 0.5f;
    if (s.n_cam > 0) {
        uint64_t k = path_index % (uint64_t)s.n_cam;
        jx = (float)s.cam_tab[2 * k];
        jy = (float)s.cam_tab[2 * k + 1];
    }
    float lx = p.top_left[0] + p.steps[0] * ((float)(c.x_offset + x) + jx);
    float ly = p.top_left[1] - p.steps[1] * ((float)(c.y_offset + y) + jy);
    origin = transform_coordinate(v3{lx, ly, -p.ppd}, p.cam_transform);
    v3 diff;
    if (p.projection == 0) {
        diff = v3{origin.x - p.cam_transform[12], origin.y - p.cam_transform[13],
                  origin.z - p.cam_transform[14]};
    } else {
        v3 f = transform_coordinate(v3{0, 0, -1}, p.cam_transform);
        diff = v3{f.x - p.cam_transform[12], f.y - p.cam_transform[13], f.z - p.cam_transform[14]};
    }
    dir = normalize(diff);
}

// arealight.go:42-56 with the table row chosen by the caller
RT_HD v3 light_sample(const DevScene &s, const PassParams &p, uint64_t idx, float &pdf) {
    double a = 0.5, b = 0.5;
    if (s.n_light > 0) {
        uint64_t k = idx % (uint64_t)s.n_light;
        a = s.light_tab[2 * k];
        b = s.light_tab[2 * k + 1];
    }
    v3 sample{(float)(a * 2 - 1) * p.light_size[0], (float)(b * 2 - 1) * p.light_size[1], 0};
    pdf = 1.0f / (4.0f * p.light_size[0] * p.light_size[1]);
    return transform_coordinate(sample, p.light_transform);
}

constexpr int kHemiMaxTries = 64;

// utility.go:50-70 on the counter RNG (bounded number of rejection tries)
RT_HD v3 random_in_hemisphere(uint64_t stream, uint32_t &ctr, const v3 &normal) {
    v3 p{0, 0, 0};
    for (int tries = 0; tries < kHemiMaxTries; ++tries) {
        p.x = rng_f32(stream, ctr++) * 2 - 1;
        p.y = rng_f32(stream, ctr++) * 2 - 1;
        p.z = rng_f32(stream, ctr++) * 2 - 1;
        if (lensqr(p) < 1) break;
        if (tries == kHemiMaxTries - 1) p = v3{p.x * 0.5f, p.y * 0.5f, p.z * 0.5f};
    }
    if (dot(p, normal) > 0.0f) return p;
    return v3{p.x * -1, p.y * -1, p.z * -1};
}

// trace.go:154-187 getMaterialParameters: Kd, optionally times the nearest
// texel of map_Kd (texture.go:48-59; no V flip, repeat wrap by uv - floor(uv)).
// `mat_word` is the w component of the triangle's tri_shade word.
RT_HD v3 material_diffuse(const DevScene &s, int32_t tri_pos, int32_t mat_word, float u, float v) {
    const int32_t mat = mat_word & (kLightBit - 1);
    v3 kd{s.kd[3 * mat], s.kd[3 * mat + 1], s.kd[3 * mat + 2]};
    if (s.mat_tex == nullptr || s.tri_uv == nullptr) return kd;
    const int32_t slot = s.mat_tex[mat];
    if (slot < 0 || s.tex_data[slot] == nullptr) return kd;
    const float *tc = s.tri_uv + 6 * (size_t)tri_pos;
    // TextureCoords[0]*(1-U-V) + TextureCoords[1]*U + TextureCoords[2]*V
    const float w0 = 1.0f - u - v;
    float tu = (tc[0] * w0 + tc[2] * u) + tc[4] * v;
    float tv = (tc[1] * w0 + tc[3] * u) + tc[5] * v;
    tu = tu - (float)floor((double)tu);
    tv = tv - (float)floor((double)tv);
    const int32_t w = s.tex_dim[2 * slot], h = s.tex_dim[2 * slot + 1];
    int32_t px = (int32_t)(tu * (float)w), py = (int32_t)(tv * (float)h);
    if (px < 0 || py < 0 || px >= w || py >= h) return v3{0, 0, 0};   // image.RGBA.At outside bounds
    const uint8_t *t = s.tex_data[slot] + 4 * ((size_t)py * (size_t)w + (size_t)px);
    // RGBA() widens 8-bit channels to 16 bits (c | c<<8), then * float32(1/65535)
    const float k = (float)(1.0 / 65535.0);
    v3 smp{(float)(t[0] * 257) * k, (float)(t[1] * 257) * k, (float)(t[2] * 257) * k};
    smp = v3{gomath::clamp01(smp.x), gomath::clamp01(smp.y), gomath::clamp01(smp.z)};
    return v3{kd.x * smp.x, kd.y * smp.y, kd.z * smp.z};
}

// trace.go:49-71: one light sample at a path vertex.  Returns true when a
// shadow ray must be cast (lightIncident < 0); colour is what the sample adds
// to `shading` if that ray reaches the light.
RT_HD bool shadow_candidate(const DevScene &s, const PassParams &p, const v3 &point,
                            const v3 &tri_normal, const v3 &diffuse, uint64_t light_idx,
                            v3 &dir_n, float &dist, v3 &colour) {
    float pdf;
    const v3 ls = light_sample(s, p, light_idx, pdf);
    const v3 sr{ls.x - point.x, ls.y - point.y, ls.z - point.z};
    dist = len(sr);
    dir_n = normalize(sr);
    const v3 ln{p.light_normal[0], p.light_normal[1], p.light_normal[2]};
    const float li = dot(dir_n, ln);
    if (!(li < 0)) return false;
    const float theta_l = (float)gomath::go_max((double)(-li), 0.0);
    const float theta = (float)gomath::go_max((double)dot(dir_n, tri_normal), 0.0);
    const float r2 = lensqr(sr);
    const float pi32 = (float)gomath::kPi;
    const float k = theta_l * theta / (r2 * pdf * pi32);
    colour = v3{(diffuse.x * p.light_emission[0]) * k, (diffuse.y * p.light_emission[1]) * k,
                (diffuse.z * p.light_emission[2]) * k};
    return true;
}

// trace.go:100-104: the BRDF factor of a bounce that did hit something
RT_HD float brdf_factor(const v3 &cur_dir, const v3 &sample, const v3 &normal) {
    const float brdf_theta = dot(cur_dir, sample) * -1;
    const float theta = dot(sample, normal);
    const double pdf = gomath::go_cos((double)brdf_theta) / gomath::kPi;
    return (float)(gomath::go_cos((double)theta) / (gomath::kPi * pdf));
}

// main.go:158-182 for one channel
RT_HD uint8_t resolve_channel(float sum, float inv_divisor, bool gamma_on, double inv_gamma) {
    float c = sum * inv_divisor;
    if (gamma_on) c = (float)gomath::go_pow((double)c, inv_gamma);
    c = gomath::clamp01(c);
    float x = 255 * c;
    if (!(x == x)) return 0;
    if (x <= 0) return 0;
    if (x >= 255) return 255;
    return (uint8_t)x;
}

}  // namespace rtc
