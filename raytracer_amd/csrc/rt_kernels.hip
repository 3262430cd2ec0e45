// rt_kernels.hip -- CDNA4 (gfx950) kernels and their launchers.
//
//   k_trace_ref   persistent-wave BVH traversal in the reference's visit order
//                 (bvh.go:117-144) with the reference's arithmetic
//                 (aabb.go:62-99, triangle.go:78-103); float32 and float64
//                 instantiations.  Waves pull rays from a global cursor and
//                 refill idle lanes (ballot + prefix popcount), so a wave never
//                 idles on a few long rays.
//   k_raygen      camera.go:99-129 -> coalesced ray buffer
//   k_shade       trace.go:44-109 for path vertex k of every live path: folds
//                 the previous vertex's shadow results, emits this vertex's
//                 shadow rays and the bounce ray into a compacted queue
//   k_finish      last vertex's shadows + the back-to-front fold (trace.go:114-118)
//   k_accumulate  main.go:133-152 pixel sums, in sample order
//   k_resolve     main.go:158-182
//
// There is no matrix work anywhere on this path: every kernel is bound by
// cache/HBM latency and bandwidth under divergent access, so the design rules
// are 16-byte loads, compact records, full occupancy and wave-level queueing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rt_core.h"
#include "rt_device.h"
#include "rt_internal.h"

using namespace rtc;

// Static device workspace of the entry point that must not allocate
// (rt_render_tile_static): one copy per GPU the code object is loaded on, held
// until the process ends.  In host code the array's name is only a handle for
// hipGetSymbolAddress; every use of its memory goes through the per-device
// address obtained once at set-up (rt::FixedWorkspace::ptr).
constexpr size_t kStaticScratchBytes = (size_t)RT_STATIC_WORKSPACE_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace rt {

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess)                                                           \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

// ---------------------------------------------------------------------------
// wave helpers (wave64)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
// Lanes of the wave for which `p` holds.  (HIP's __ballot takes an int: a bool
// argument costs a select to 0/1 and a compare back to a mask on every call.)
__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Wave-aggregated queue append: one atomic per wave, slots in lane order.
__device__ __forceinline__ uint32_t wave_append(uint32_t *counter, bool want) {
    const uint64_t m = wave_ballot(want);
    if (m == 0) return 0;
    const int leader = __ffsll((unsigned long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(m & lanemask_lt());
}

// Block-aggregated queue append: ONE atomic per workgroup, slots in thread
// order.  Every thread of the block must call it, the same number of times;
// `call` is that running count (alternates the LDS slot so consecutive calls
// need no third barrier).
constexpr int kShadeBlock = 512;
__device__ __forceinline__ uint32_t block_append(uint32_t *counter, bool want, int call) {
    __shared__ uint32_t s_cnt[2][kShadeBlock / 64];
    __shared__ uint32_t s_base[2];
    const int slot = call & 1;
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t m = wave_ballot(want);
    if (lane_id() == 0) s_cnt[slot][wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t i = 0; i < blockDim.x / 64; ++i) total += s_cnt[slot][i];
        s_base[slot] = total ? atomicAdd(counter, total) : 0;
    }
    __syncthreads();
    uint32_t base = s_base[slot];
    for (uint32_t i = 0; i < wave; ++i) base += s_cnt[slot][i];
    return base + (uint32_t)__popcll(m & lanemask_lt());
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// ---------------------------------------------------------------------------
// Ray hand-out for the persistent traversal waves.  A returning atomic on one
// word completes every ~11 ns chip-wide, so a wave must not take rays a few at
// a time from the global cursor: it reserves a chunk of kRayChunk consecutive
// rays with one atomic, keeps the chunk's bounds in scalar registers and fills
// its idle lanes from it without touching memory again until it runs dry.
// ---------------------------------------------------------------------------
constexpr uint32_t kRayChunk = 256;

struct RayFeed {
    uint32_t pos = 0, end = 0;   // wave-uniform: the reserved chunk [pos, end)
    bool exhausted = false;      // the global cursor has passed the last ray
};

// GUIDED instantiations of the traversal kernels (the launcher picks them for launches
// of a few million rays) take smaller chunks towards the end of the launch: a wave whose
// last chunk ended within one round of chunks (one chunk for every wave of the grid) of
// the last ray asks for half a chunk, within half a round for a quarter.  The rays a
// wave holds when the cursor runs out are what it finishes alone, at falling lane counts;
// in a small launch that tail is most of the launch (a single batch of 2 M primary rays:
// 1.30 instead of 1.76 ms in the reference-order kernel, 0.73 instead of 0.81 ms in the
// ordered one), in a launch of 100 M rays it is nothing.  Large launches run the plain
// instantiation, whose code is untouched: the same logic chosen by a kernel argument
// cost them 0.5 % (one more scalar argument and a branch in kernels this close to their
// limit), small chunks throughout cost 20 % (the cursor's atomic has a rate).
template <bool GUIDED>
__device__ __forceinline__ uint32_t feed_chunk(const RayFeed &f, uint32_t n) {
    if (!GUIDED) return kRayChunk;
    const uint32_t round = gridDim.x * (blockDim.x >> 6) * kRayChunk;
    const uint32_t left = n > f.end ? n - f.end : 0u;   // the cursor is at or past this wave's last chunk
    return left < round / 2 ? kRayChunk / 4 : (left < round ? kRayChunk / 2 : kRayChunk);
}

// Called by all lanes of the wave.  `idle` = ballot of lanes without a ray, `is_idle` = this lane's own bit of it.
// Returns this lane's new ray index, or -1.
template <bool GUIDED = false>
__device__ __forceinline__ int32_t feed_take(RayFeed &f, uint32_t *cursor, uint32_t n, uint64_t idle, bool is_idle) {
    if (f.pos >= f.end && !f.exhausted) {
        const uint32_t want = feed_chunk<GUIDED>(f, n);
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(cursor, want);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n) {
            f.exhausted = true;
        } else {
            f.pos = base;
            f.end = (n - base < want) ? n : base + want;
        }
    }
    const uint32_t avail = f.end - f.pos;
    // rank among the idle lanes: one mbcnt pair (a 64-bit "lanes below me" mask takes a
    // shift, a subtraction, two ands and two bit counts)
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
    const int32_t mine = (is_idle && rank < avail) ? (int32_t)(f.pos + rank) : -1;
    const uint32_t n_idle = (uint32_t)__popcll(idle);
    f.pos += n_idle < avail ? n_idle : avail;
    return mine;
}

// ---------------------------------------------------------------------------
// Traversal, reference order.  One lane = one ray; the wave is persistent.
// ---------------------------------------------------------------------------

// TOP: the tree is read through its linked view (DevScene::linked), whose
// first n_top nodes -- the kTopNodes nodes with the largest boxes, which
// link_nodes moves to the front of the array: a parent's box is never smaller
// than its child's, so they form a top of the tree, and area is how likely a ray
// is to reach a node -- are copied into LDS by every block: the kernel is bound
// by the number of per-lane global loads.  A copy is shared by the 16 waves of a
// 1024-thread block (80 KB, two blocks per CU fill its 160 KB of LDS); with
// 256-thread blocks the same LDS held eight copies of a fifth as many nodes.
// Visit order and arithmetic are the same.
constexpr int kTopNodes = 2560;
constexpr int kTopBlock = 1024;

template <typename Real, bool COUNT, bool TOP, bool GUIDED = false>
__global__ void __launch_bounds__(TOP ? kTopBlock : 256)
k_trace_ref(DevScene s, const rt_ray *__restrict__ rays, rt_hit *__restrict__ hits,
            uint32_t n_host, const uint32_t *__restrict__ n_dev, uint32_t *next,
            unsigned long long *counters, int refill_below, int descend_min_full, int leaf_min_full) {
    __shared__ f4 top[TOP ? 2 * kTopNodes : 1];
    if (TOP) {
        for (int i = threadIdx.x; i < 2 * s.n_top; i += blockDim.x) top[i] = s.linked[i];
        __syncthreads();
    }
    const uint32_t top_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)top;   // the table's offset in LDS, which the hand-written ds_read takes
    const uint32_t n = n_dev ? *n_dev : n_host;
    int32_t idx = -1;            // ray owned by this lane, -1 = idle
    RayFeed feed;
    Ray<Real> r;
    Hit<Real> h;
    // Where the lane stands, in ONE register so that every vote of the wave is one
    // compare: a node index below n_nodes -- it walks, that node is tested next; a
    // negative value -- it holds a passed leaf (leaf_a, leaf_n: triangles left; nxt:
    // the node to go on with afterwards); an index past the last node -- its ray is
    // finished, or it has none (then idx is -1 as well).
    int32_t cur = 0x7fffffff, nxt = 0, leaf_a = 0, leaf_n = 0;
    uint32_t c_nodes = 0, c_box = 0, c_fetch = 0, c_test = 0;

    for (;;) {
        // retire rays that walked past the last node, refill idle lanes
        if (idx >= 0 && cur >= s.n_nodes) {
            f4 out;
            out.x = i2f(h.pos);
            out.y = (float)h.t;
            out.z = (float)h.u;
            out.w = (float)h.v;
            *reinterpret_cast<f4 *>(hits + idx) = out;
            idx = -1;
        }
        uint64_t active = wave_ballot(idx >= 0);
        if (!(feed.exhausted && feed.pos >= feed.end) && (int)__popcll(active) < refill_below) {
            const int32_t mine = feed_take<GUIDED>(feed, next, n, ~active, idx < 0);
            if (mine >= 0) {
                const f4 *rp = reinterpret_cast<const f4 *>(rays + mine);
                const f4 a = rp[0], b = rp[1];
                ray_init(r, a.x, a.y, a.z, b.x, b.y, b.z);
                h.t = (Real)a.w;
                h.u = 0;
                h.v = 0;
                h.pos = -1;
                cur = 0;
                idx = mine;
            }
            active = wave_ballot(idx >= 0);
        }
        if (active == 0) {
            if (feed.exhausted && feed.pos >= feed.end) break;
            continue;   // the chunk ran dry in the middle of a refill: take the next one
        }
        // GUIDED instantiations: the phase thresholds scale with the lanes alive once the wave
        // has shrunk (see k_trace_fast)
        int descend_min = descend_min_full, leaf_min = leaf_min_full;
        if (GUIDED) {
            const int n_alive = (int)__popcll(active);
            if (n_alive < 48) {
                descend_min = (n_alive * descend_min_full + 63) >> 6;
                leaf_min = (n_alive * leaf_min_full + 63) >> 6;
            }
        }

        // phase A: walk boxes in the reference's order until this lane stands on a
        // passed leaf; the phase ends once fewer than descend_min lanes still walk
        // while others wait on a leaf
        for (;;) {
            const bool walking = (uint32_t)cur < (uint32_t)s.n_nodes;
            const uint64_t m_walk = wave_ballot(walking);
            if (m_walk == 0) break;
            if ((int)__popcll(m_walk) < descend_min && wave_ballot(cur < 0) != 0) break;
            if (walking) {
            const int32_t node = cur;
            f4 lo, hi;
            if (TOP) {
                // Lanes standing on a node of the LDS table read it from there, the others
                // from memory, into the SAME registers, and one wait covers both: the two
                // groups of lanes are disjoint, and a load writes only the lanes that issued
                // it.  (Written as two C++ branches, the second waits for the first one's
                // data before it issues, and the record is moved between registers.)
                typedef float v4f __attribute__((ext_vector_type(4)));
                v4f wlo, whi;
                asm volatile("" : "=v"(wlo), "=v"(whi));
                if (node < s.n_top) {
                    const uint32_t at = top_base + ((uint32_t)node << 5);
                    asm volatile("ds_read_b128 %0, %2\n\t"
                                 "ds_read_b128 %1, %2 offset:16"
                                 : "+v"(wlo), "+v"(whi) : "v"(at));
                } else {
                    // 32-bit byte offset beside the scalar base (a linked view has fewer than 2^25 nodes)
                    const uint32_t at = (uint32_t)node << 5;
                    asm volatile("global_load_dwordx4 %0, %2, %3\n\t"
                                 "global_load_dwordx4 %1, %2, %3 offset:16"
                                 : "+v"(wlo), "+v"(whi) : "v"(at), "s"(s.linked));
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(wlo), "+v"(whi));
                lo = f4{wlo.x, wlo.y, wlo.z, wlo.w};
                hi = f4{whi.x, whi.y, whi.z, whi.w};
            } else {
                lo = s.nodes[2 * node];
                hi = s.nodes[2 * node + 1];
                // keep the record two 16-byte loads: left alone the compiler splits it into
                // four 8-byte ones around the root test below, and the kernel is bound by the
                // number of per-lane loads
                asm volatile("" : "+v"(lo.x), "+v"(lo.y), "+v"(lo.z), "+v"(lo.w));
                asm volatile("" : "+v"(hi.x), "+v"(hi.y), "+v"(hi.z), "+v"(hi.w));
            }
            const int32_t a = f2i(lo.w), b = f2i(hi.w);
            bool pass = true;
            if (node != 0) {
                Real t0, t1;
                if (COUNT) c_box++;
                pass = slab(r, lo, hi, t0, t1) && t0 < h.t && t1 > 0;
            }
            if (TOP) {
                // a = next node when this one fails or is a leaf; b = leaf range or ~left child
                if (b >= 0) {
                    cur = a;
                    if (pass && (b >> 24) > 0) {
                        leaf_a = b & 0xFFFFFF;
                        leaf_n = b >> 24;
                        nxt = a;
                        cur = -1;
                    }
                } else {
                    if (COUNT && pass) c_nodes++;
                    cur = pass ? -1 - b : a;
                }
            } else if (b >= 0) {
                cur = node + 1;
                if (pass && b > 0) {
                    leaf_a = a;
                    leaf_n = b;
                    nxt = node + 1;
                    cur = -1;
                }
            } else {
                if (COUNT && pass) c_nodes++;
                cur = pass ? node + 1 : a;
            }
            }
        }
        // phase B: the leaf's triangles, in array order; ends when no lane holds a
        // leaf, or when fewer than leaf_min do while enough lanes wait to walk
        for (;;) {
            const bool in_leaf = cur < 0;
            const uint64_t m_leaf = wave_ballot(in_leaf);
            if (m_leaf == 0) break;
            if ((int)__popcll(m_leaf) < leaf_min &&
                (int)__popcll(wave_ballot((uint32_t)cur < (uint32_t)s.n_nodes)) >= descend_min) break;
            if (in_leaf) {
            // two triangles per step, fetched together so that their loads overlap,
            // tested in array order
            const bool two = leaf_n > 1;
            const f4 *qa, *qb;
            if (TOP) {
                // 32-bit byte offsets beside the scalar base (triangle positions of a linked view are below 2^24)
                const uint32_t off_a = __umul24((uint32_t)leaf_a, 48u);   // a 24-bit multiply is full rate, the 32-bit one a quarter
                uint32_t off_b = off_a + (two ? 48u : 0u);
                asm volatile("" : "+v"(off_b));
                qa = reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(s.tris) + off_a);
                qb = reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(s.tris) + off_b);
            } else {
                qa = s.tris + 3 * (size_t)leaf_a;
                qb = qa + (two ? 3 : 0);
            }
            const f4 a0 = qa[0], a1 = qa[1], a2 = qa[2];
            const f4 b0 = qb[0], b1 = qb[1], b2 = qb[2];
            if (COUNT) c_fetch += two ? 2 : 1;
            bool tested = tri_test(r, a0, a1, a2, leaf_a, h);
            if (COUNT && tested) c_test++;
            if (two) {
                tested = tri_test(r, b0, b1, b2, leaf_a + 1, h);
                if (COUNT && tested) c_test++;
            }
            leaf_a += 2;
            leaf_n -= 2;
            if (leaf_n <= 0) cur = nxt;
            }
        }
    }
    if (COUNT) {
        c_nodes = wave_sum(c_nodes);
        c_box = wave_sum(c_box);
        c_fetch = wave_sum(c_fetch);
        c_test = wave_sum(c_test);
        if (lane_id() == 0) {
            atomicAdd(counters + 1, (unsigned long long)c_nodes);
            atomicAdd(counters + 2, (unsigned long long)c_box);
            atomicAdd(counters + 3, (unsigned long long)c_fetch);
            atomicAdd(counters + 4, (unsigned long long)c_test);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(counters + 0, (unsigned long long)n);
    }
}

// ---------------------------------------------------------------------------
// Traversal, fast mode, over the 4-wide view of the tree: a step decodes the four
// child boxes of one 64-byte record, runs the reference's slab arithmetic on
// them (the decoded boxes contain the reference's), enters the nearest child
// that is hit and leaves the others on a per-lane stack, farthest first, so most
// of the tree behind the first hit is never fetched.  The stack's top kStackLds
// entries live in LDS ([entry][lane], 8 bytes each: child reference + entry
// distance), deeper entries spill to scratch.  A popped entry whose entry
// distance lies beyond the current hit distance is dropped without touching
// memory; one at the hit distance or within kTieSlack of it is kept, like a
// child box entered there, because a triangle in it may tie with the current
// hit and sit earlier in the array, and exact ties go to the earlier position
// (the triangle the reference's left-first order meets first).  Triangle
// arithmetic is the reference's.
//
// The loop alternates two phases, "descend through interior records until this
// lane holds a leaf" and "test the leaf's triangles", so that the lanes of a
// wave run the same code most of the time.  Results equal the reference-order
// traversal except where the reference's own box test prunes a triangle it
// would accept.
// ---------------------------------------------------------------------------
constexpr int kStackTotal = kFastStackEntries;
// Records of the top of the tree that every block of the default ordered kernel
// keeps in LDS (k_trace_fast: kTop): with the 8-entry stack, 16 KB + 80 x 80
// bytes per block, seven blocks fill a CU's 160 KB -- the occupancy the kernel
// has without the table.  Larger tables at lower occupancy measured slower.
constexpr int kFastTopRecords = 80;
typedef float f2 __attribute__((ext_vector_type(2)));

// The three plane pairs (near, far) of one child box, from plane offsets to slab
// distances: decode with one fused multiply-add per pair (offset * step + frame
// origin), then the reference's (bound - origin) * invDir.  Every instruction
// works on a (near, far) pair and broadcasts one half of a register pair that
// holds two different scalars, selected by op_sel.  The trailing wait state is
// the one the compiler puts between a packed result and an unpacked reader.
__device__ __forceinline__ void slab_planes(f2 &tx, f2 &ty, f2 &tz, f2 f_oxy, f2 f_ozsx, f2 f_syz,
                                            f2 r_oxy, f2 r_ozix, f2 r_iyz) {
    asm("v_pk_fma_f32 %0, %0, %4, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 %1, %1, %5, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %2, %2, %5, %4 op_sel:[0,1,0] op_sel_hi:[1,1,0]\n\t"
        "v_pk_add_f32 %0, %0, %6 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %1, %6 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %2, %2, %7 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %0, %0, %7 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %1, %1, %8 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "s_nop 0"
        : "+v"(tx), "+v"(ty), "+v"(tz)
        : "v"(f_oxy), "v"(f_ozsx), "v"(f_syz), "v"(r_oxy), "v"(r_ozix), "v"(r_iyz));
}

// The near-origin form (rt_core.h: kNearOriginDiagonals): the plane offsets are
// decoded relative to the ray, offset * step + (frame origin - ray origin), with
// that difference formed once per record, so the per-child subtraction is gone.
// d_xy = (origin - o).xy; d_zsx = ((origin - o).z, step x).
__device__ __forceinline__ void slab_planes_near(f2 &tx, f2 &ty, f2 &tz, f2 d_xy, f2 d_zsx, f2 f_syz,
                                                 f2 r_ozix, f2 r_iyz) {
    asm("v_pk_fma_f32 %0, %0, %4, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 %1, %1, %5, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %2, %2, %5, %4 op_sel:[0,1,0] op_sel_hi:[1,1,0]\n\t"
        "v_pk_mul_f32 %0, %0, %6 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %1, %1, %7 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %2, %2, %7 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "s_nop 0"
        : "+v"(tx), "+v"(ty), "+v"(tz)
        : "v"(d_xy), "v"(d_zsx), "v"(f_syz), "v"(r_ozix), "v"(r_iyz));
}

template <bool COUNT, int kStackLds, bool NEAR = false, int kTop = 0, bool GUIDED = false>
__global__ void __launch_bounds__(256)
k_trace_fast(DevScene s, const rt_ray *__restrict__ rays, rt_hit *__restrict__ hits,
             uint32_t n_host, const uint32_t *__restrict__ n_dev, uint32_t *next,
             unsigned long long *counters, int refill_below, int descend_min_full, int leaf_min_full) {
    constexpr int kStackSpill = kStackTotal - kStackLds;
    constexpr int kBlock = 256;
    __shared__ uint2 lds_stack[kBlock * kStackLds];
    // Values that no descent step reads -- the hit's u and v, the ray's index and
    // its direction (the descent uses the inverse direction) -- live in LDS,
    // [value][thread], when that lets a SIMD hold one more wave (the descent step
    // sets the register count).  With the record table below the LDS is what limits
    // the waves, seven per SIMD either way, and the values stay in registers.
    constexpr bool kParkRegs = kTop > 0;
    __shared__ float park[kParkRegs ? 1 : 6 * kBlock];
    float *my_park = park + (kParkRegs ? 0 : threadIdx.x);
    float park_reg[6] = {0, 0, 0, 0, 0, 0};
#define RT_PARK_SET(k, v) do { if (kParkRegs) park_reg[k] = (v); else my_park[(k) * kBlock] = (v); } while (0)
#define RT_PARK_GET(k) (kParkRegs ? park_reg[k] : my_park[(k) * kBlock])
    // The first kTop records of the array -- those with the largest frames, a top
    // of the tree (flatten.cpp: top_records_first) -- are copied into LDS by every
    // block: the kernel is bound by the number of per-lane global loads (one
    // cache-tag lookup per lane and 16-byte load, one lookup per clock and CU), and
    // this table serves most record fetches.  A record takes 80 bytes there, so that
    // the lanes of one LDS read, standing on different records, spread over all banks.
    constexpr int kTopStride = 5;    // f4 per record in the LDS table
    __shared__ f4 top[kTop ? kTopStride * kTop : 1];
    const uint32_t n_top = kTop ? (uint32_t)(s.n_wide < kTop ? s.n_wide : kTop) : 0u;
    const uint32_t top_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)top;   // the table's offset in LDS, which the hand-written ds_read takes
    if (kTop) {
        for (uint32_t i = threadIdx.x; i < 4u * n_top; i += kBlock) top[(i >> 2) * kTopStride + (i & 3u)] = s.wide[i];
        __syncthreads();
    }
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t lane = lane_id();
    // The LDS part of the stack is laid out [entry][thread of the block], and a lane's
    // stack pointer is kept as the BYTE offset of its next free entry from the start of
    // that array (entry x block size x 8 + thread x 8): a push or pop then needs no address
    // arithmetic beyond one add, and "is this entry in LDS" is one compare with a
    // constant for every lane.
    constexpr uint32_t kEntryStride = kBlock * sizeof(uint2);    // bytes between a lane's consecutive entries
    constexpr uint32_t kLdsLimit = kStackLds * kEntryStride;     // offsets below this are LDS entries
    const uint32_t sp_empty = threadIdx.x * (uint32_t)sizeof(uint2);
    char *const stack_bytes = reinterpret_cast<char *>(lds_stack);
    uint2 spill[kStackSpill];
    RayFeed feed;
    // origin and inverse direction, as the register pairs the packed slab step reads:
    // (origin x, origin y), (origin z, 1/dir x), (1/dir y, 1/dir z)
    f2 r_oxy{0, 0}, r_ozix{0, 0}, r_iyz{0, 0};
    float h_t = 0;               // closest hit so far
    float h_lim = 0;             // h_t * kTieSlack, kept finite: entry distances up to here are still entered
    int32_t h_pos = -1;
    // The lane's whole state is where it stands, in the form of a child reference:
    // an interior record index (>= 0 as a signed number), or leaf bit | triangles
    // left << 24 | next triangle (below -2), or one of two values no leaf can
    // have: kRefDone (-1), its ray is finished and waits to be written out, and
    // kLaneIdle (-2), it holds no ray.  Every vote of the wave is then ONE compare
    // of this register.
    constexpr uint32_t kLaneIdle = 0xFFFFFFFEu;
    uint32_t where = kLaneIdle;
    uint32_t sp = sp_empty;      // byte offset of the next free stack entry (see above)
    uint32_t c_nodes = 0, c_box = 0, c_fetch = 0, c_test = 0, c_wa = 0, c_wb = 0, c_lb = 0;

    auto push = [&](uint32_t ref, float tn) {
        const uint2 e = make_uint2(ref, (uint32_t)f2i(tn));
        if (sp < kLdsLimit) *reinterpret_cast<uint2 *>(stack_bytes + sp) = e;
        else if (sp < kStackTotal * kEntryStride) spill[sp / kEntryStride - kStackLds] = e;
        sp += kEntryStride;   // deeper than both levels cannot happen: a wide tree whose depth x 3 exceeds the stack is not built
    };
    // Stands the lane on `ref`: an interior record, or a leaf.
    auto stand_on = [&](uint32_t ref) { where = ref; };
    // Next stacked entry that can still hold a closer hit; kRefDone when the ray
    // is finished.  The LDS slot is read unconditionally (index clamped) and replaced
    // from the spill array only when the entry lives there, so that both accesses
    // keep their own address space.
    auto pop_next = [&]() {
        while (sp >= kEntryStride) {
            sp -= kEntryStride;
            uint2 e = *reinterpret_cast<const uint2 *>(stack_bytes + (sp < kLdsLimit ? sp : sp_empty));
            asm volatile("" : "+v"(e.x), "+v"(e.y));   // the LDS read stays an LDS read
            if (sp >= kLdsLimit) e = spill[sp / kEntryStride - kStackLds];
            if (i2f((int32_t)e.y) <= h_lim) {
                stand_on(e.x);
                return;
            }
        }
        where = kRefDone;
    };

    for (;;) {
        // retire finished rays, refill idle lanes
        if (where == kRefDone) {
            f4 out;
            out.x = i2f(h_pos);
            out.y = h_t;
            out.z = RT_PARK_GET(0);
            out.w = RT_PARK_GET(1);
            const int32_t idx = f2i(RT_PARK_GET(2));
            *reinterpret_cast<f4 *>(hits + idx) = out;
            where = kLaneIdle;
        }
        uint64_t active = wave_ballot(where != kLaneIdle);
        if (!(feed.exhausted && feed.pos >= feed.end) && (int)__popcll(active) < refill_below) {
            const int32_t mine = feed_take<GUIDED>(feed, next, n, ~active, where == kLaneIdle);
            if (mine >= 0) {
                const f4 *rp = reinterpret_cast<const f4 *>(rays + mine);
                const f4 a = rp[0], b = rp[1];
                r_oxy.x = a.x; r_oxy.y = a.y; r_ozix.x = a.z;
                float inv_x, inv_y, inv_z;
                recip3<float>(b.x, b.y, b.z, inv_x, inv_y, inv_z);   // ray.go:33-37
                r_ozix.y = inv_x; r_iyz.x = inv_y; r_iyz.y = inv_z;
                h_t = a.w;
                h_lim = fminf(a.w * kTieSlack, kF32Max);
                h_pos = -1;
                RT_PARK_SET(0, 0.0f);
                RT_PARK_SET(1, 0.0f);
                RT_PARK_SET(2, i2f(mine));
                RT_PARK_SET(3, b.x);
                RT_PARK_SET(4, b.y);
                RT_PARK_SET(5, b.z);
                sp = sp_empty;
                stand_on(s.fast_root);
            }
            active = wave_ballot(where != kLaneIdle);
        }
        if (active == 0) {
            if (feed.exhausted && feed.pos >= feed.end) break;
            continue;   // the chunk ran dry in the middle of a refill: take the next one
        }
        // GUIDED instantiations (small launches, which are mostly drain): the two phase
        // thresholds are shares of a FULL wave; once the queue has run dry and the wave has
        // shrunk, fixed numbers would end the descent phase at the first lane that reaches a
        // leaf and never end the leaf phase early, so they scale with the lanes alive.
        int descend_min = descend_min_full, leaf_min = leaf_min_full;
        if (GUIDED) {
            const int n_alive = (int)__popcll(active);
            if (n_alive < 48) {
                descend_min = (n_alive * descend_min_full + 63) >> 6;
                leaf_min = (n_alive * leaf_min_full + 63) >> 6;
            }
        }

        // phase A: lanes standing on an interior record descend.  The phase ends
        // once fewer than descend_min lanes still descend while others wait on a
        // leaf: one long descent must not idle the wave.
        for (;;) {
            const bool descending = (int32_t)where >= 0;
            const uint64_t m_desc = wave_ballot(descending);
            if (m_desc == 0) break;
            if ((int)__popcll(m_desc) < descend_min && wave_ballot((int32_t)where < -2) != 0) break;
            if (descending) {
            // byte offset in 32 bits (the record table stays below 4 GB: at most one record
            // per triangle, 2^24 of them), so that the loads take the table's base from
            // scalar registers and no 64-bit address is built per lane
            f4 q0, q1, q2, q3;
            if (kTop) {
                // Lanes standing on a record of the LDS table read it from there, the others
                // from memory, into the SAME registers, and one wait covers both: the two
                // groups of lanes are disjoint, and a load writes only the lanes that issued
                // it.  Written as two branches in C++, the compiler either merges them into
                // loads through a generic pointer or makes the second branch wait for the
                // first one's data before it issues; hence the instructions are given here.
                typedef float v4f __attribute__((ext_vector_type(4)));
                v4f w0, w1, w2, w3;
                asm volatile("" : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3));
                if (where < n_top) {
                    const uint32_t at = top_base + __umul24(where, 16u * kTopStride);
                    asm volatile("ds_read_b128 %0, %4\n\t"
                                 "ds_read_b128 %1, %4 offset:16\n\t"
                                 "ds_read_b128 %2, %4 offset:32\n\t"
                                 "ds_read_b128 %3, %4 offset:48"
                                 : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(at));
                } else {
                    const uint32_t at = where << 6;
                    asm volatile("global_load_dwordx4 %0, %4, %5\n\t"
                                 "global_load_dwordx4 %1, %4, %5 offset:16\n\t"
                                 "global_load_dwordx4 %2, %4, %5 offset:32\n\t"
                                 "global_load_dwordx4 %3, %4, %5 offset:48"
                                 : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(at), "s"(s.wide));
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
                q0 = f4{w0.x, w0.y, w0.z, w0.w}; q1 = f4{w1.x, w1.y, w1.z, w1.w};
                q2 = f4{w2.x, w2.y, w2.z, w2.w}; q3 = f4{w3.x, w3.y, w3.z, w3.w};
            } else {
                const f4 *q = reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(s.wide) + (where << 6));
                q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
            }
            if (COUNT) { c_nodes++; c_box += 4; if ((wave_ballot(true) & lanemask_lt()) == 0) c_wa++; }
            // frame of the record, then per child: decode the six planes with one
            // fused multiply-add each (byte -> float conversions are single
            // instructions), and the reference's slab arithmetic
            // (bound - origin) * invDir on the decoded planes, two planes per
            // packed instruction.
            // The plane a ray meets first on an axis is the lower one when it
            // travels up that axis and the upper one otherwise: pick the near and
            // far plane words once per record (the sign of the inverse direction
            // is the direction's, also for +-0), so that no per-child min/max of
            // the two slab distances is needed
            const bool nx = r_ozix.y < 0, ny = r_iyz.x < 0, nz = r_iyz.y < 0;
            const uint32_t lox = (uint32_t)f2i(q1.z), loy = (uint32_t)f2i(q1.w), loz = (uint32_t)f2i(q2.x);
            const uint32_t hix = (uint32_t)f2i(q2.y), hiy = (uint32_t)f2i(q2.z), hiz = (uint32_t)f2i(q2.w);
            const uint32_t plx = nx ? hix : lox, ply = ny ? hiy : loy, plz = nz ? hiz : loz;   // near
            const uint32_t phx = nx ? lox : hix, phy = ny ? loy : hiy, phz = nz ? loz : hiz;   // far
            // Packed operands are register pairs as they were loaded; which half an
            // instruction broadcasts is said by its op_sel bits, so no pair has to
            // be assembled with moves: frame origin x|y = (q0.x,q0.y), origin z and
            // step x = (q0.z,q0.w), step y|z = (q1.x,q1.y); the ray likewise
            // (origin x|y), (origin z, 1/dir x), (1/dir y|z).
            const f2 f_oxy{q0.x, q0.y}, f_ozsx{q0.z, q0.w}, f_syz{q1.x, q1.y};
            // near-origin form: the frame origin relative to the ray, once per record
            const f2 d_xy{f_oxy.x - r_oxy.x, f_oxy.y - r_oxy.y}, d_zsx{f_ozsx.x - r_ozix.x, f_ozsx.y};
            float key[4];
            uint32_t ref[4] = {(uint32_t)f2i(q3.x), (uint32_t)f2i(q3.y), (uint32_t)f2i(q3.z), (uint32_t)f2i(q3.w)};
#define RT_BYTE(v, k) ((float)(((v) >> (8 * (k))) & 0xFFu))
#define RT_SLAB(k)                                                                          \
            {                                                                               \
                f2 tx{RT_BYTE(plx, k), RT_BYTE(phx, k)};                                    \
                f2 ty{RT_BYTE(ply, k), RT_BYTE(phy, k)};                                    \
                f2 tz{RT_BYTE(plz, k), RT_BYTE(phz, k)};                                    \
                if (NEAR) slab_planes_near(tx, ty, tz, d_xy, d_zsx, f_syz, r_ozix, r_iyz);  \
                else slab_planes(tx, ty, tz, f_oxy, f_ozsx, f_syz, r_oxy, r_ozix, r_iyz);   \
                const float tn = fmaxf(fmaxf(tx.x, ty.x), tz.x);                            \
                const float tf = fminf(fminf(tx.y, ty.y), tz.y);                            \
                /* the box is met (unused slot: inverted box); whether it lies within reach of */ \
                /* the hit distance is asked once per child after the sort, where the old      */ \
                /* "was it met" compare stood, so the step has no min(tf, h_lim) per child     */ \
                const bool hit = (tn <= tf) & (tf > 0);                                     \
                key[k] = hit ? tn : kF32Inf;                                                \
            }
            RT_SLAB(0)
            RT_SLAB(1)
            RT_SLAB(2)
            RT_SLAB(3)
#undef RT_SLAB
#undef RT_BYTE
            wide_sort4(key, ref);
            // missed children carry +inf and sort last; h_lim is finite, so one compare says
            // both "met" and "entry distance within reach of the hit distance"
            if (!(key[0] <= h_lim)) {
                pop_next();
            } else {
                // farthest first, so that the nearest of the rest is popped first
                if (key[3] <= h_lim) push(ref[3], key[3]);
                if (key[2] <= h_lim) push(ref[2], key[2]);
                if (key[1] <= h_lim) push(ref[1], key[1]);
                stand_on(ref[0]);
            }
            }
        }
        // phase B: lanes holding a leaf test its triangles in array order, two
        // fetched at a time so that their loads overlap; a lane that finishes its
        // leaf takes its next stack entry at once.  The phase ends when no lane
        // holds a leaf, or when fewer than leaf_min do while enough lanes wait to
        // descend (the few keep their place in the leaf for the next round).
        for (;;) {
            const bool in_leaf = (int32_t)where < -2;
            const uint64_t m_leaf = wave_ballot(in_leaf);
            if (m_leaf == 0) break;
            if ((int)__popcll(m_leaf) < leaf_min &&
                (int)__popcll(wave_ballot((int32_t)where >= 0)) >= descend_min) break;
            if (in_leaf) {
            const int32_t tri_i = (int32_t)(where & 0xFFFFFFu), tri_n = (int32_t)((where >> 24) & 0x7Fu);
            const bool two = tri_n > 1;
            // 32-bit byte offsets (at most 2^24 triangles of 48 bytes), see the descent step
            const uint32_t off_a = (uint32_t)tri_i * 48u;
            // a lane with one triangle left reads the array's first triangle as its second:
            // all such lanes of the wave then share one cache lookup per load
            uint32_t off_b = two ? off_a + 48u : 0u;
            asm volatile("" : "+v"(off_b));   // stays one 32-bit offset beside the scalar base
            const f4 *qa = reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(s.tris) + off_a);
            const f4 *qb = reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(s.tris) + off_b);
            f4 a0 = qa[0], a1 = qa[1], a2 = qa[2];
            const f4 b0 = qb[0], b1 = qb[1], b2 = qb[2];
            // all six loads are issued here, together: left alone the compiler moves
            // the first triangle's into the branch that tests it and splits them up
            asm volatile("" :: "v"(a0.x), "v"(a1.x), "v"(a2.x));
            if (COUNT) { c_fetch += two ? 2 : (tri_n > 0 ? 1 : 0); c_lb++; if ((wave_ballot(true) & lanemask_lt()) == 0) c_wb++; }
            // the full ray only exists here
            Ray<float> r;
            r.ox = r_oxy.x; r.oy = r_oxy.y; r.oz = r_ozix.x;
            r.dx = RT_PARK_GET(3); r.dy = RT_PARK_GET(4); r.dz = RT_PARK_GET(5);
            r.ix = r_ozix.y; r.iy = r_iyz.x; r.iz = r_iyz.y;
            // both triangles are evaluated side by side without early exits, then
            // offered to the hit in array order (a triangle is accepted below the
            // current distance, or at it from an earlier position).  A leaf of no
            // triangles is what an unused slot refers to; its inverted box lets a ray
            // in only when rounding collapses the box's extent.
            float ta, ua, va, tb, ub, vb;
            bool in_a, in_b;
            const bool faced_a = tri_eval(r, a0, a1, a2, ta, ua, va, in_a);
            const bool faced_b = tri_eval(r, b0, b1, b2, tb, ub, vb, in_b);
            if (COUNT) c_test += (tri_n > 0 && faced_a ? 1u : 0u) + (two && faced_b ? 1u : 0u);
            const bool take_a = (tri_n > 0) & in_a & ((ta < h_t) | ((ta == h_t) & (tri_i < h_pos)));
            h_t = take_a ? ta : h_t;
            h_pos = take_a ? tri_i : h_pos;
            const bool take_b = two & in_b & ((tb < h_t) | ((tb == h_t) & (tri_i + 1 < h_pos)));
            h_t = take_b ? tb : h_t;
            h_pos = take_b ? tri_i + 1 : h_pos;
            h_lim = fminf(h_t * kTieSlack, kF32Max);
            if (take_a | take_b) {
                RT_PARK_SET(0, take_b ? ub : ua);
                RT_PARK_SET(1, take_b ? vb : va);
            }
            if (tri_n <= 2) {
                pop_next();
            } else {
                where = kLeafBit | ((uint32_t)(tri_n - 2) << 24) | (uint32_t)(tri_i + 2);
            }
            }
        }
    }
    if (COUNT) {
        c_nodes = wave_sum(c_nodes);
        c_box = wave_sum(c_box);
        c_fetch = wave_sum(c_fetch);
        c_test = wave_sum(c_test);
        c_wa = wave_sum(c_wa);
        c_wb = wave_sum(c_wb);
        c_lb = wave_sum(c_lb);
        if (lane == 0) {
            atomicAdd(counters + 5, (unsigned long long)c_wa);
            atomicAdd(counters + 6, (unsigned long long)c_wb);
            atomicAdd(counters + 7, (unsigned long long)c_lb);
            atomicAdd(counters + 1, (unsigned long long)c_nodes);
            atomicAdd(counters + 2, (unsigned long long)c_box);
            atomicAdd(counters + 3, (unsigned long long)c_fetch);
            atomicAdd(counters + 4, (unsigned long long)c_test);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(counters + 0, (unsigned long long)n);
    }
#undef RT_PARK_SET
#undef RT_PARK_GET
}

// ---------------------------------------------------------------------------
// Path-tracing wavefront kernels
// ---------------------------------------------------------------------------
enum : uint32_t { kDone = 0, kActive = 1, kLastVertex = 2, kComplete = 3 };

__device__ __forceinline__ uint32_t pack_head(uint32_t state, uint32_t n_terms, uint32_t ctr) {
    return state | (n_terms << 2) | (ctr << 11);
}
__device__ __forceinline__ uint32_t head_state(uint32_t packed) { return packed & 3u; }
__device__ __forceinline__ uint32_t head_terms(uint32_t packed) { return (packed >> 2) & 0x1FFu; }
__device__ __forceinline__ uint32_t head_ctr(uint32_t packed) { return packed >> 11; }

struct Batch {
    uint32_t P;                // paths in this batch = n_pix * n_samp
    uint32_t n_pix;            // pixels in this batch
    uint32_t pix0;             // first pixel (tile-linear)
    uint32_t s0;               // first sample index
    uint32_t n_samp;
};

__global__ void __launch_bounds__(256)
k_raygen(DevScene s, PassParams p, Batch b, PathBuffers w) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id == 0) w.counts[0] = b.P;
    if (id >= b.P) return;
    const uint32_t pl = id % b.n_pix, sl = id / b.n_pix;
    const PathCtx ctx = locate_path(p, b.pix0 + pl, b.s0 + sl);
    v3 o, d;
    camera_ray(s, p, ctx, o, d);
    f4 *rp = reinterpret_cast<f4 *>(w.rays[0] + id);
    rp[0] = f4{o.x, o.y, o.z, kF32Max};
    rp[1] = f4{d.x, d.y, d.z, 0.0f};
    const uint64_t stream = rng_stream(ctx.rng_seed, ctx.path_index);
    // the primary ray of path `id` sits in slot `id` of queue 0
    w.head[id] = PathHead{(uint32_t)stream, (uint32_t)(stream >> 32), pack_head(kActive, 0, 0), id};
    if (w.path_rays) w.path_rays[id] = 1;   // the primary ray
}

// Sum of the shadow samples that reached the light, scaled and clamped
// (trace.go:63-77).  `in` is the queue the shadow rays of the previous vertex
// were traced in.  The reference's test is "T >= lightDistance || the hit
// triangle is the light" on a cast that started with tmin = lightDistance
// (trace.go:59-63): a triangle is only ever accepted below the current tmin, so
// T >= lightDistance holds exactly when no triangle was accepted, and the ray's
// length need not be kept.
__device__ __forceinline__ v3 fold_shadows(const DevScene &s, const PassParams &p,
                                           const PathBuffers &w, int in, uint32_t id, size_t P) {
    v3 shading{0, 0, 0};
    for (int i = 0; i < p.L; ++i) {
        const f4 slot = w.slots[(size_t)i * P + id];
        const int32_t q = f2i(slot.w);
        if (q < 0) continue;
        const rt_hit sh = w.hits[in][q];
        const bool reached = sh.t < kF32Max;   // rayCast returned a result
        const bool lit = reached && (sh.tri_pos < 0 || (f2i(s.tri_shade[sh.tri_pos].w) & kLightBit) != 0);
        if (lit) {
            shading.x = shading.x + slot.x;
            shading.y = shading.y + slot.y;
            shading.z = shading.z + slot.z;
        }
    }
    const float invl = 1 / (float)p.L;
    return v3{gomath::clamp01(shading.x * invl), gomath::clamp01(shading.y * invl),
              gomath::clamp01(shading.z * invl)};
}

// One try of RandomInUnitSphere (utility.go:34-48) on the counter RNG: the point
// made of draws c, c + 1, c + 2 of a stream.
__device__ __forceinline__ v3 sphere_try(uint64_t stream, uint32_t c) {
    v3 p;
    p.x = rng_f32(stream, c) * 2 - 1;
    p.y = rng_f32(stream, c + 1) * 2 - 1;
    p.z = rng_f32(stream, c + 2) * 2 - 1;
    return p;
}

// random_in_hemisphere's rejection loop (rt_core.h) for the lanes of a wave whose
// `need` is set, with the tries shared out over the whole wave.  A try costs three
// 64-bit hash draws and succeeds with probability pi/6: run lane by lane, a lane is
// done after 1.9 tries on average, but the wave waits for its unluckiest lane
// (about 6.5 tries).  The RNG is counter based, so try j of any path can be
// evaluated by any lane: after one try of its own per lane, every further round
// deals the still open paths' next tries to ALL 64 lanes (open path of rank r gets
// the lanes r, r + n, r + 2n, ... when n paths are open) and each path takes its
// first accepted try in try order.  Result and draw count per path are exactly
// those of the sequential loop: the first try j with |p|^2 < 1, 3 (j + 1) draws
// consumed; try 63 is taken in any case, halved (the loop's bound).
// Every lane of the wave must call this (the exec mask must be full).
__device__ __forceinline__ v3 wave_random_in_unit_sphere(bool need, uint64_t stream, uint32_t &ctr) {
    const uint32_t lane = lane_id();
    const uint32_t c0 = ctr;
    v3 p = sphere_try(stream, c0);
    bool open = need && !(lensqr(p) < 1);
    uint32_t used = 1;          // tries the lane's own path has consumed once it is closed
    uint32_t tried = 1;         // wave-uniform: tries evaluated so far for every open path
    uint64_t pending = wave_ballot(open);
    while (pending != 0) {
        const uint32_t n = (uint32_t)__popcll(pending);
        uint32_t m = 64u / n;                                   // tries per open path this round
        if (m > (uint32_t)kHemiMaxTries - tried) m = (uint32_t)kHemiMaxTries - tried;
        // the open path of rank r announces its lane to lane r (the other lanes fill the rest
        // of the permutation, so that every destination is written exactly once)
        const uint32_t rank = (uint32_t)__popcll(pending & lanemask_lt());
        const uint32_t dest = open ? rank : n + (uint32_t)__popcll(~pending & lanemask_lt());
        const int lane_of_rank = __builtin_amdgcn_ds_permute((int)(dest << 2), (int)lane);
        // this lane works for rank r = lane mod n on try `tried + lane / n`; (lane + 1/2) / n
        // keeps the approximate reciprocal's error away from the integers
        const uint32_t q = (uint32_t)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)n));
        const uint32_t r = lane - q * n;
        const int target = __builtin_amdgcn_ds_bpermute((int)(r << 2), lane_of_rank) << 2;
        const uint32_t t_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(target, (int)(uint32_t)stream);
        const uint32_t t_hi = (uint32_t)__builtin_amdgcn_ds_bpermute(target, (int)(uint32_t)(stream >> 32));
        const uint32_t t_c0 = (uint32_t)__builtin_amdgcn_ds_bpermute(target, (int)c0);
        const uint32_t j = tried + q;
        v3 w = sphere_try(((uint64_t)t_hi << 32) | t_lo, t_c0 + 3u * j);
        bool ok = lensqr(w) < 1;
        if (j == (uint32_t)kHemiMaxTries - 1 && !ok) {          // the bounded loop's last try is taken, halved
            w = v3{w.x * 0.5f, w.y * 0.5f, w.z * 0.5f};
            ok = true;
        }
        const uint64_t got = wave_ballot(q < m && ok);
        // the tries of the path of rank r sit in lanes r, r + n, ..., r + (m - 1) n, in try order
        uint64_t every_nth = 0;
        for (uint32_t i = 0; i < m; ++i) every_nth |= 1ull << (n * i);     // wave-uniform
        const uint64_t mine = (got >> rank) & every_nth;
        const bool closes = open && mine != 0;
        const int src = (int)((closes ? rank + (uint32_t)__ffsll((unsigned long long)mine) - 1u : lane) << 2);
        const float wx = i2f(__builtin_amdgcn_ds_bpermute(src, f2i(w.x)));
        const float wy = i2f(__builtin_amdgcn_ds_bpermute(src, f2i(w.y)));
        const float wz = i2f(__builtin_amdgcn_ds_bpermute(src, f2i(w.z)));
        const uint32_t wj = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)j);
        if (closes) {
            p = v3{wx, wy, wz};
            used = wj + 1;
            open = false;
        }
        tried += m;
        pending = wave_ballot(open);
    }
    if (need) ctr = c0 + 3u * used;
    return p;
}

// Path vertex k.  Reads queue (k & 1), writes queue ((k+1) & 1).
__global__ void __launch_bounds__(kShadeBlock) __attribute__((amdgpu_waves_per_eu(6, 6)))
k_shade(DevScene s, PassParams p, Batch b, PathBuffers w, int k) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t P = b.P;
    PathHead hd{0, 0, 0, 0};
    if (id < b.P) hd = w.head[id];
    const bool live = id < b.P && head_state(hd.packed) == kActive;
    const int in = k & 1, out = in ^ 1;
    uint32_t *out_count = w.counts + k + 1;

    bool at_vertex = false;
    rt_hit hit{};
    v3 ro{0, 0, 0}, rd{0, 0, 0}, prev_shading{0, 0, 0};
    uint32_t ctr = head_ctr(hd.packed);
    if (live) {
        if (k > 0) prev_shading = fold_shadows(s, p, w, in, id, P);
        hit = w.hits[in][hd.bounce_q];
        if (!(hit.t < kF32Max)) {
            // primary miss: Trace returns the background; bounce miss: the path ends
            // with k terms, the last vertex's shading is what was just folded
            w.terms_a[(size_t)k * P + id] = f4{prev_shading.x, prev_shading.y, prev_shading.z, 0.0f};
            reinterpret_cast<uint2 *>(w.head + id)[1] = make_uint2(pack_head(kComplete, (uint32_t)k, ctr), 0u);
        } else {
            const f4 *rp = reinterpret_cast<const f4 *>(w.rays[in] + hd.bounce_q);
            const f4 a = rp[0], c = rp[1];
            ro = v3{a.x, a.y, a.z};
            rd = v3{c.x, c.y, c.z};
            at_vertex = true;
        }
    }

    v3 point{0, 0, 0}, normal{0, 0, 1}, diffuse{0, 0, 0};
    if (at_vertex) {
        point = v3{ro.x + rd.x * hit.t, ro.y + rd.y * hit.t, ro.z + rd.z * hit.t};
        const f4 q0 = s.tri_shade[hit.tri_pos];
        normal = v3{q0.x, q0.y, q0.z};
        diffuse = material_diffuse(s, hit.tri_pos, f2i(q0.w), hit.u, hit.v);
    }
    const uint64_t stream = (uint64_t)hd.stream_lo | ((uint64_t)hd.stream_hi << 32);
    const uint64_t lbase = light_base(stream);

    // shadow rays of this vertex (the loop is uniform over the wave: L is a pass constant)
    uint32_t cast_here = 0;
    for (int i = 0; i < p.L; ++i) {
        v3 dn{0, 0, 0}, col{0, 0, 0};
        float dist = 0;
        bool cast = false;
        if (at_vertex)
            cast = shadow_candidate(s, p, point, normal, diffuse,
                                    lbase + (uint64_t)k * (uint64_t)p.L + (uint64_t)i, dn, dist, col);
        const uint32_t q = block_append(out_count, cast, i);
        cast_here += cast ? 1u : 0u;
        if (at_vertex) {
            if (cast) {
                f4 *rp = reinterpret_cast<f4 *>(w.rays[out] + q);
                rp[0] = f4{point.x, point.y, point.z, dist};
                rp[1] = f4{dn.x, dn.y, dn.z, 0.0f};
            }
            w.slots[(size_t)i * P + id] = f4{col.x, col.y, col.z, i2f(cast ? (int32_t)q : -1)};
        }
    }

    // bounce (trace.go:81-108)
    const bool bounce = at_vertex && k < p.B;
    if (at_vertex && !bounce) {
        w.terms_a[(size_t)k * P + id] = f4{prev_shading.x, prev_shading.y, prev_shading.z, 0.0f};
        reinterpret_cast<uint2 *>(w.head + id)[1] = make_uint2(pack_head(kLastVertex, (uint32_t)k + 1, ctr), 0u);
    }
    // utility.go:50-70: a point in the unit ball, flipped into the normal's hemisphere
    // (every lane takes part in the draw, see wave_random_in_unit_sphere)
    v3 sample = wave_random_in_unit_sphere(bounce, stream, ctr);
    if (bounce) {
        if (!(dot(sample, normal) > 0.0f)) sample = v3{sample.x * -1, sample.y * -1, sample.z * -1};
        sample = normalize(sample);
    }
    const uint32_t bq = block_append(out_count, bounce, p.L);
    if (w.path_rays && at_vertex) w.path_rays[id] += cast_here + (bounce ? 1u : 0u);
    if (bounce) {
        f4 *rp = reinterpret_cast<f4 *>(w.rays[out] + bq);
        rp[0] = f4{point.x, point.y, point.z, kF32Max};
        rp[1] = f4{sample.x, sample.y, sample.z, 0.0f};
        // the direction the path arrived with is the traced ray's own (trace.go:95 currentRay)
        const float f = brdf_factor(rd, sample, normal);
        // the term counts only if the bounce ray hits something: a path whose bounce
        // misses ends with this vertex as its last, whose BRDF term is zero
        w.terms_a[(size_t)k * P + id] = f4{prev_shading.x, prev_shading.y, prev_shading.z, diffuse.x * f};
        w.terms_b[(size_t)k * P + id] = Pair32{diffuse.y * f, diffuse.z * f};
        reinterpret_cast<uint2 *>(w.head + id)[1] = make_uint2(pack_head(kActive, 0, ctr), bq);
    }
}

// Last vertex's shadows, then E = shading[0] + brdf[0] folded back to front
// (trace.go:114-118): x = 0 for the last vertex, x = brdf[t] * (shading[t+1] + x)
// going down, E = shading[0] + x.
__global__ void __launch_bounds__(256)
k_finish(DevScene s, PassParams p, Batch b, PathBuffers w, int last_queue) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= b.P) return;
    const size_t P = b.P;
    const PathHead hd = w.head[id];
    const uint32_t st = head_state(hd.packed);
    const int n = (int)head_terms(hd.packed);
    v3 rgb{0.0f / 255.0f, 0.0f / 255.0f, 0.0f / 255.0f};
    if (n > 0) {
        // shading term of the last vertex: its shadow rays were traced last, or
        // (bounce missed) the vertex after it already folded and stored them
        v3 s_hi;
        if (st == kLastVertex) {
            s_hi = fold_shadows(s, p, w, last_queue, id, P);
        } else {
            const f4 a = w.terms_a[(size_t)n * P + id];
            s_hi = v3{a.x, a.y, a.z};
        }
        v3 x{0.0f, 0.0f, 0.0f};
        v3 s_pending{0, 0, 0};                       // shading[t] while the loop works on t
        if (n > 1) {
            const f4 a = w.terms_a[(size_t)(n - 1) * P + id];
            s_pending = v3{a.x, a.y, a.z};
        }
        for (int t = n - 2; t >= 0; --t) {
            const f4 a = w.terms_a[(size_t)t * P + id];   // (shading[t-1], brdf[t].x)
            const Pair32 yz = w.terms_b[(size_t)t * P + id];
            x = v3{a.w * (s_hi.x + x.x), yz.x * (s_hi.y + x.y), yz.y * (s_hi.z + x.z)};
            s_hi = s_pending;
            s_pending = v3{a.x, a.y, a.z};
        }
        rgb = v3{s_hi.x + x.x, s_hi.y + x.y, s_hi.z + x.z};
    }
    w.radiance[id] = rgb.x;
    w.radiance[P + id] = rgb.y;
    w.radiance[2 * P + id] = rgb.z;
}

// pixelColor += rayColor, samples in order (main.go:137-151)
__global__ void __launch_bounds__(256)
k_accumulate(Batch b, const float *__restrict__ radiance, float *__restrict__ accum) {
    const uint32_t pl = blockIdx.x * blockDim.x + threadIdx.x;
    if (pl >= b.n_pix) return;
    const size_t pix = (size_t)b.pix0 + pl;
    const size_t P = b.P;
    float r = accum[3 * pix], g = accum[3 * pix + 1], bl = accum[3 * pix + 2];
    for (uint32_t sidx = 0; sidx < b.n_samp; ++sidx) {
        const size_t i = (size_t)sidx * b.n_pix + pl;
        r = r + radiance[i];
        g = g + radiance[P + i];
        bl = bl + radiance[2 * P + i];
    }
    accum[3 * pix] = r;
    accum[3 * pix + 1] = g;
    accum[3 * pix + 2] = bl;
}

__global__ void __launch_bounds__(256)
k_resolve(const float *__restrict__ accum, size_t pixels, float inv_divisor, int gamma_on,
          double inv_gamma, uint8_t *__restrict__ rgba) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pixels) return;
    uchar4 o;
    o.x = resolve_channel(accum[3 * i], inv_divisor, gamma_on != 0, inv_gamma);
    o.y = resolve_channel(accum[3 * i + 1], inv_divisor, gamma_on != 0, inv_gamma);
    o.z = resolve_channel(accum[3 * i + 2], inv_divisor, gamma_on != 0, inv_gamma);
    o.w = 255;
    reinterpret_cast<uchar4 *>(rgba)[i] = o;
}

// Rays of the batch's paths, summed per progress bin (see ProgressBins): one
// LDS atomic per path, one global atomic per bin and block.
__global__ void __launch_bounds__(256)
k_progress_bins(PassParams p, Batch b, const uint32_t *__restrict__ path_rays, uint32_t step, int n_bins,
                unsigned long long *bins) {
    __shared__ unsigned int part[kProgressBins];
    if (threadIdx.x < kProgressBins) part[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < b.P) {
        const uint32_t pl = id % b.n_pix, sl = id / b.n_pix;
        const PathCtx ctx = locate_path(p, b.pix0 + pl, b.s0 + sl);
        uint64_t bin = ctx.path_index / step;
        if (bin > (uint64_t)(n_bins - 1)) bin = (uint64_t)(n_bins - 1);
        atomicAdd(&part[bin], path_rays[id]);
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)n_bins && part[threadIdx.x])
        atomicAdd(bins + threadIdx.x, (unsigned long long)part[threadIdx.x]);
}

__global__ void k_sum_counts(const uint32_t *counts, int n, unsigned long long *total) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < n; ++i) t += counts[i];
        *total += t;
    }
}

// ---------------------------------------------------------------------------
// Device scene
// ---------------------------------------------------------------------------

// A caller-provided block of device memory that render_core uses as its path
// workspace instead of the scene's allocated one.
struct FixedWorkspace {
    void *ptr = nullptr;          // device address of the block
    size_t bytes = 0;
    hipEvent_t done = nullptr;    // recorded at the end of every call that used it
    std::mutex mu;                // calls take turns on the host
    std::atomic<bool> ready{false};   // set once under g_static_mu, read by every static call
};

// Per-device state of the static workspace (the `scratch` array at the top of
// this file).
constexpr int kMaxDevices = 64;
static FixedWorkspace g_static[kMaxDevices];
static std::mutex g_static_mu;

// Once per device, with that device current (scene creation is the library's
// per-device set-up call).
static rt_status static_scratch_setup(int device) {
    if (device < 0 || device >= kMaxDevices) return RT_OK;   // such a device simply has no static workspace
    std::lock_guard<std::mutex> lock(g_static_mu);
    FixedWorkspace &ds = g_static[device];
    if (ds.ready) return RT_OK;
    void *p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(RT_ERR_DEVICE, "hipGetSymbolAddress returned a null address for the static workspace");
    HIP_TRY(hipEventCreateWithFlags(&ds.done, hipEventDisableTiming));
    ds.ptr = p;
    ds.bytes = kStaticScratchBytes;
    ds.ready = true;
    return RT_OK;
}

namespace {

template <typename T>
rt_status upload(T **dst, const void *src, size_t bytes) {
    *dst = nullptr;
    if (bytes == 0) bytes = 16;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(dst), bytes));
    if (src) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return RT_OK;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) ok = false;
        if (ok && prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

// Linked form of the reference-order node array (DevScene::linked): the max_top
// nodes with the largest boxes (the root first) are moved to the front, the
// others keep their preorder behind them, and the implicit
// "next node = this one + 1" of the preorder array becomes an explicit link.
// Returns false when a leaf does not fit the packed encoding (more than 127
// triangles, or a first triangle beyond 2^24): such trees use the plain array.
static bool link_nodes(const Bvh &bvh, const std::vector<float> &nodes, int32_t max_top,
                       std::vector<float> &out, int32_t &n_top) {
    const int32_t nn = bvh.n_nodes;
    if (nn <= 0 || (int64_t)nn >= (1 << 27)) return false;   // the kernel addresses the records with 32-bit byte offsets
    // the max_top nodes with the largest boxes: the chance that a ray reaches a node
    // grows with its box's area, and a parent's box is never smaller than its
    // child's, so these are the most fetched nodes and form a top of the tree
    std::vector<float> area((size_t)nn);
    for (int32_t i = 0; i < nn; ++i) {
        const float *q = &nodes[(size_t)i * 8];
        const float d0 = q[4] - q[0], d1 = q[5] - q[1], d2 = q[6] - q[2];
        const float a = d0 * d1 + d0 * d2 + d1 * d2;
        // empty leaves (NaN or inverted boxes) last
        area[(size_t)i] = (d0 >= 0 && d1 >= 0 && d2 >= 0 && std::isfinite(a)) ? a : -1.0f;
    }
    area[0] = std::numeric_limits<float>::infinity();   // the kernel knows the root as node 0
    n_top = nn < max_top ? nn : max_top;
    std::vector<int32_t> by_area((size_t)nn);
    for (int32_t i = 0; i < nn; ++i) by_area[(size_t)i] = i;
    std::partial_sort(by_area.begin(), by_area.begin() + n_top, by_area.end(), [&](int32_t x, int32_t y) {
        return area[(size_t)x] != area[(size_t)y] ? area[(size_t)x] > area[(size_t)y] : x < y;
    });
    std::vector<uint8_t> front((size_t)nn, 0);
    for (int32_t k = 0; k < n_top; ++k) front[(size_t)by_area[(size_t)k]] = 1;
    std::vector<int32_t> to_new((size_t)nn + 1);
    int32_t next_top = 0, next_deep = n_top;
    for (int32_t i = 0; i < nn; ++i) to_new[(size_t)i] = front[(size_t)i] ? next_top++ : next_deep++;
    to_new[(size_t)nn] = nn;               // "past the last node" stays the end marker
    out.assign((size_t)nn * 8, 0.0f);
    for (int32_t i = 0; i < nn; ++i) {
        const float *src = &nodes[(size_t)i * 8];
        float *dst = &out[(size_t)to_new[(size_t)i] * 8];
        std::memcpy(dst, src, 32);
        const int32_t a = rtc::f2i(src[3]), b = rtc::f2i(src[7]);
        if (b >= 0) {                      // leaf: (first, count) -> next link + packed range
            if (b > 127 || a < 0 || a >= (1 << 24)) return false;
            dst[3] = rtc::i2f(to_new[(size_t)i + 1]);
            dst[7] = rtc::i2f((b << 24) | a);
        } else {                           // interior: skip link + left child
            if (a < 0 || a > nn) return false;
            dst[3] = rtc::i2f(to_new[(size_t)a]);
            dst[7] = rtc::i2f(-1 - to_new[(size_t)i + 1]);
        }
    }
    return true;
}

rt_status scene_create(const rt_mesh &mesh, const Bvh &bvh, int device, Scene **out) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(RT_ERR_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device < 0 || device >= ndev)
        return fail(RT_ERR_INVALID, "device index " + std::to_string(device) + " out of range");
    if (bvh.n_tris != mesh.n_tris)
        return fail(RT_ERR_INVALID, "BVH was built for a different triangle count");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    {
        const rt_status ss = static_scratch_setup(device);
        if (ss != RT_OK) return ss;
    }

    // an error return below releases whatever was uploaded so far
    struct Release {
        void operator()(Scene *p) const { scene_destroy(p); }
    };
    std::unique_ptr<Scene, Release> sc(new Scene);
    sc->device = device;
    sc->n_tris = mesh.n_tris;
    sc->n_nodes = bvh.n_nodes;
    sc->max_depth = bvh.max_depth;
    sc->n_materials = mesh.n_materials;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    sc->num_cus = prop.multiProcessorCount;

    FlatScene flat;
    rt_status st = flatten(mesh, bvh, flat);
    if (st != RT_OK) return st;
    find_light(mesh, sc->light);

    if ((st = upload(&sc->d_nodes, flat.nodes.data(), flat.nodes.size() * 4)) != RT_OK) return st;
    {
        std::vector<float> linked;
        int32_t n_top = 0;
        if (link_nodes(bvh, flat.nodes, kTopNodes, linked, n_top)) {
            if ((st = upload(&sc->d_linked, linked.data(), linked.size() * 4)) != RT_OK) return st;
            sc->n_top = n_top;
        }
    }
    if ((st = upload(&sc->d_tris, flat.tris.data(), flat.tris.size() * 4)) != RT_OK) return st;
    {
        std::vector<float> shade((size_t)mesh.n_tris * 4);
        for (size_t i = 0; i < (size_t)mesh.n_tris; ++i) {
            std::memcpy(&shade[4 * i], &flat.tris[12 * i], 12);   // the normal leads the triangle record
            shade[4 * i + 3] = rtc::i2f(flat.tri_mat[i]);
        }
        if ((st = upload(&sc->d_tri_shade, shade.data(), shade.size() * 4)) != RT_OK) return st;
    }
    if ((st = upload(&sc->d_kd, mesh.kd, (size_t)mesh.n_materials * 12)) != RT_OK) return st;
    if (mesh.texcoords)
        if ((st = upload(&sc->d_tri_uv, flat.tri_uv.data(), flat.tri_uv.size() * 4)) != RT_OK) return st;
    sc->mat_tex.assign((size_t)mesh.n_materials, -1);
    if (mesh.texture)
        for (int32_t m = 0; m < mesh.n_materials; ++m) sc->mat_tex[(size_t)m] = mesh.texture[m];
    if ((st = upload(&sc->d_mat_tex, sc->mat_tex.data(), sc->mat_tex.size() * 4)) != RT_OK) return st;
    sc->node_bytes = flat.nodes.size() * 4;
    sc->tri_bytes = flat.tris.size() * 4;
    if (flat.fast_ok) {
        if ((st = upload(&sc->d_wide, flat.wide.data(), flat.wide.size() * 4)) != RT_OK) return st;
        sc->fast_root = flat.fast_root;
        sc->n_wide = (int32_t)(flat.wide.size() / rtc::kWideFloats);
        sc->wide_bytes = flat.wide.size() * 4;
        std::memcpy(sc->near_lo, flat.near_lo, sizeof sc->near_lo);
        std::memcpy(sc->near_hi, flat.near_hi, sizeof sc->near_hi);
    }
    if ((st = upload(&sc->d_cursors, nullptr, (size_t)kCursorSlots * kCursorStride * sizeof(uint32_t))) != RT_OK) return st;
    HIP_TRY(hipMemset(sc->d_cursors, 0, (size_t)kCursorSlots * kCursorStride * sizeof(uint32_t)));
    HIP_TRY(hipEventCreateWithFlags(&sc->ws_done, hipEventDisableTiming));

    // default sample tables (same row counts as the reference's)
    std::vector<double> cam((size_t)12345 * 2), lig((size_t)100001 * 2);
    default_sample_table(12345, 0xC0FFEE01ULL, cam.data());
    default_sample_table(100001, 0xC0FFEE02ULL, lig.data());
    if ((st = scene_set_tables(*sc, cam.data(), 12345, lig.data(), 100001)) != RT_OK) return st;
    *out = sc.release();
    return RT_OK;
}

rt_status scene_set_tables(Scene &sc, const double *cam, int32_t n_cam, const double *lig,
                           int32_t n_lig) {
    if (n_cam < 0 || n_lig < 0 || (n_cam > 0 && !cam) || (n_lig > 0 && !lig))
        return fail(RT_ERR_INVALID, "bad sample table arguments");
    DeviceGuard guard(sc.device);
    if (sc.d_cam_tab) (void)hipFree(sc.d_cam_tab);
    if (sc.d_light_tab) (void)hipFree(sc.d_light_tab);
    sc.d_cam_tab = sc.d_light_tab = nullptr;
    rt_status st;
    if ((st = upload(&sc.d_cam_tab, cam, (size_t)n_cam * 16)) != RT_OK) return st;
    if ((st = upload(&sc.d_light_tab, lig, (size_t)n_lig * 16)) != RT_OK) return st;
    sc.n_cam = n_cam;
    sc.n_light = n_lig;
    return RT_OK;
}

rt_status scene_set_texture(Scene &sc, int32_t slot, const uint8_t *rgba, int32_t w, int32_t h) {
    if (slot < 0 || slot >= 4096 || !rgba || w <= 0 || h <= 0)
        return fail(RT_ERR_INVALID, "bad texture arguments");
    DeviceGuard guard(sc.device);
    if ((size_t)slot >= sc.tex_ptrs.size()) {
        sc.tex_ptrs.resize((size_t)slot + 1, nullptr);
        sc.tex_dims.resize(2 * ((size_t)slot + 1), 0);
    }
    if (sc.tex_ptrs[(size_t)slot]) (void)hipFree(sc.tex_ptrs[(size_t)slot]);
    sc.tex_ptrs[(size_t)slot] = nullptr;
    uint8_t *d = nullptr;
    rt_status st;
    if ((st = upload(&d, rgba, (size_t)w * (size_t)h * 4)) != RT_OK) return st;
    sc.tex_ptrs[(size_t)slot] = d;
    sc.tex_dims[2 * (size_t)slot] = w;
    sc.tex_dims[2 * (size_t)slot + 1] = h;
    if (sc.d_tex_ptrs) (void)hipFree(sc.d_tex_ptrs);
    if (sc.d_tex_dims) (void)hipFree(sc.d_tex_dims);
    sc.d_tex_ptrs = nullptr;
    sc.d_tex_dims = nullptr;
    if ((st = upload(&sc.d_tex_ptrs, sc.tex_ptrs.data(), sc.tex_ptrs.size() * sizeof(uint8_t *))) != RT_OK) return st;
    if ((st = upload(&sc.d_tex_dims, sc.tex_dims.data(), sc.tex_dims.size() * 4)) != RT_OK) return st;
    // the kernels index the slot tables with the material's texture id: ids
    // without an image so far read as "no texture"
    std::vector<int32_t> usable(sc.mat_tex);
    for (int32_t &id : usable)
        if (id < 0 || (size_t)id >= sc.tex_ptrs.size() || !sc.tex_ptrs[(size_t)id]) id = -1;
    if (!usable.empty())
        HIP_TRY(hipMemcpy(sc.d_mat_tex, usable.data(), usable.size() * 4, hipMemcpyHostToDevice));
    return RT_OK;
}

void workspace_free(Scene &sc) {
    if (sc.ws_block) (void)hipFree(sc.ws_block);
    sc.ws_block = nullptr;
    sc.ws_block_bytes = 0;
    sc.ws_paths = 0;
}

rt_status scene_set_workspace_limit(Scene &sc, size_t max_bytes) {
    std::lock_guard<std::mutex> lock(sc.render_mu);
    sc.ws_byte_limit = max_bytes;
    if (max_bytes && sc.ws_block_bytes > max_bytes) {
        DeviceGuard guard(sc.device);
        if (!guard.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
        // the last call that used the workspace may still be running on its stream
        if (sc.ws_done) HIP_TRY(hipEventSynchronize(sc.ws_done));
        workspace_free(sc);
    }
    return RT_OK;
}

void scene_destroy(Scene *sc) {
    if (!sc) return;
    DeviceGuard guard(sc->device);
    workspace_free(*sc);
    for (uint8_t *p : sc->tex_ptrs)
        if (p) (void)hipFree(p);
    void *ptrs[] = {sc->d_linked, sc->d_wide, sc->d_tiles, sc->d_nodes, sc->d_tris, sc->d_tri_shade, sc->d_kd, sc->d_tri_uv, sc->d_mat_tex,
                    sc->d_tex_ptrs, sc->d_tex_dims, sc->d_cam_tab, sc->d_light_tab, sc->d_cursors};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (sc->ws_done) (void)hipEventDestroy(sc->ws_done);
    delete sc;
}

static DevScene dev_view(const Scene &sc) {
    DevScene d{};
    d.nodes = sc.d_nodes;
    d.linked = sc.d_linked;
    d.n_top = sc.n_top;
    d.tris = sc.d_tris;
    d.tri_shade = sc.d_tri_shade;
    d.kd = sc.d_kd;
    d.tri_uv = sc.d_tri_uv;
    d.mat_tex = sc.d_tex_ptrs ? sc.d_mat_tex : nullptr;
    d.tex_data = sc.d_tex_ptrs;
    d.tex_dim = sc.d_tex_dims;
    d.cam_tab = sc.d_cam_tab;
    d.light_tab = sc.d_light_tab;
    d.n_nodes = sc.n_nodes;
    d.n_tris = sc.n_tris;
    d.n_cam = sc.n_cam;
    d.n_light = sc.n_light;
    d.wide = sc.d_wide;
    d.fast_root = sc.fast_root;
    d.fast_ok = sc.d_wide ? 1 : 0;
    d.n_wide = sc.n_wide;
    return d;
}

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------

// Zeroes 32-bit words with a kernel of its own.  The render and trace calls do
// not use hipMemsetAsync for their counters and cursors: a call may be captured
// into a HIP graph, and on this runtime a captured memset of a few bytes was not
// carried out again when the graph was replayed (the ray cursor kept its value
// and the second replay traced nothing), whereas kernel nodes replay as issued.
__global__ void __launch_bounds__(256)
k_zero_words(uint32_t *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = 0u;
}

static rt_status zero_words(void *p, size_t bytes, hipStream_t stream) {
    const size_t n = bytes / 4;
    if (n == 0) return RT_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_zero_words, dim3((uint32_t)blocks), dim3(256), 0, stream, static_cast<uint32_t *>(p), n);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// Persistent grid: enough 256-thread blocks to fill every CU at the kernel's
// occupancy, never more blocks than there are 64-ray chunks.
static uint32_t trace_grid(const Scene &sc, size_t n) {
    const size_t chunks = (n + 255) / 256;
    size_t blocks = (size_t)sc.num_cus * (size_t)sc.tune_blocks_per_cu;
    if (blocks > chunks) blocks = chunks;
    if (blocks == 0) blocks = 1;
    return (uint32_t)blocks;
}

static uint32_t *next_cursor(Scene &sc) {
    const uint32_t slot = sc.cursor_next.fetch_add(1) % kCursorSlots;
    return sc.d_cursors + (size_t)slot * kCursorStride;
}

rt_status launch_trace(Scene &sc, const rt_ray *d_rays, size_t n_host, const uint32_t *d_n,
                       rt_hit *d_hits, int mode, unsigned long long *d_counters,
                       hipStream_t stream, uint32_t *d_cursor) {
    if (n_host == 0) return RT_OK;
    if (n_host > 0x7fffffffULL) return fail(RT_ERR_INVALID, "more than 2^31-1 rays in one call");
    // the cursor reset and the launch must go to the scene's device whatever device
    // the calling thread has current (`stream` must belong to the scene's device)
    DeviceGuard guard(sc.device);
    if (!guard.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    uint32_t *cursor = d_cursor ? d_cursor : next_cursor(sc);
    { const rt_status zs = zero_words(cursor, sizeof(uint32_t), stream); if (zs != RT_OK) return zs; }
    const DevScene d = dev_view(sc);
    dim3 grid(trace_grid(sc, n_host)), block(256);
    const uint32_t n32 = (uint32_t)n_host;
    // wave scheduling thresholds (refill, descend_min, leaf_min): measured optimum per kernel
    const bool near_form = mode == RT_MODE_FAST_NEAR_F32;
    if (near_form) mode = RT_MODE_FAST_F32;
    const bool ordered = mode == RT_MODE_FAST_F32 && sc.d_wide;
    const int t_refill = sc.tune_refill > 0 ? sc.tune_refill : (ordered ? 58 : 56);
    const int t_desc = sc.tune_descend_min > 0 ? sc.tune_descend_min : (ordered ? 36 : 28);
    const int t_leaf = sc.tune_leaf_min > 0 ? sc.tune_leaf_min : (ordered ? 12 : 8);
    // launches of 1 M to 32 M ray slots run the GUIDED instantiation (smaller chunks towards the end, feed_chunk);
    // below that a launch is a few chunks per wave anyway and four times the atomics show (the 97 k-sample tiles of
    // the Cornell preset: 18.1 -> 20.7 ms per frame)
    const bool guided = n_host >= ((size_t)1 << 20) && n_host <= ((size_t)1 << 25);
    switch (mode) {
    case RT_MODE_FAST_F32:
        if (sc.d_wide) {
            const int rb = t_refill;
            if (d_counters) {
                if (near_form) hipLaunchKernelGGL((k_trace_fast<true, 16, true>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, rb, t_desc, t_leaf);
                else hipLaunchKernelGGL((k_trace_fast<true, 16>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, rb, t_desc, t_leaf);
            } else if (near_form) {
                switch (sc.tune_stack) {
                case 12: hipLaunchKernelGGL((k_trace_fast<false, 12, true>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, rb, t_desc, t_leaf); break;
                case 16: hipLaunchKernelGGL((k_trace_fast<false, 16, true>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, rb, t_desc, t_leaf); break;
                case 24: hipLaunchKernelGGL((k_trace_fast<false, 24, true>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, rb, t_desc, t_leaf); break;
                default:
                    if (guided) hipLaunchKernelGGL((k_trace_fast<false, 8, true, kFastTopRecords, true>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, rb, t_desc, t_leaf);
                    else hipLaunchKernelGGL((k_trace_fast<false, 8, true, kFastTopRecords>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, rb, t_desc, t_leaf);
                    break;
                }
            } else {
                switch (sc.tune_stack) {
                case 8:
                    if (guided) hipLaunchKernelGGL((k_trace_fast<false, 8, false, kFastTopRecords, true>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, rb, t_desc, t_leaf);
                    else hipLaunchKernelGGL((k_trace_fast<false, 8, false, kFastTopRecords>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, rb, t_desc, t_leaf);
                    break;
                case 12: hipLaunchKernelGGL((k_trace_fast<false, 12>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, rb, t_desc, t_leaf); break;
                case 24: hipLaunchKernelGGL((k_trace_fast<false, 24>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, rb, t_desc, t_leaf); break;
                default: hipLaunchKernelGGL((k_trace_fast<false, 16>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, rb, t_desc, t_leaf); break;
                }
            }
            break;
        }
        // trees that do not fit the ordered-traversal encoding run the reference order
        [[fallthrough]];
    case RT_MODE_STRICT_F32:
        if (d.linked) { block = dim3(kTopBlock); grid = dim3((grid.x * 256 + kTopBlock - 1) / kTopBlock); }
        if (d_counters)
            { if (d.linked) hipLaunchKernelGGL((k_trace_ref<float, true, true>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, t_refill, t_desc, t_leaf); else hipLaunchKernelGGL((k_trace_ref<float, true, false>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, t_refill, t_desc, t_leaf); }
        else
            { if (d.linked && guided) hipLaunchKernelGGL((k_trace_ref<float, false, true, true>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, t_refill, t_desc, t_leaf); else if (d.linked) hipLaunchKernelGGL((k_trace_ref<float, false, true>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, t_refill, t_desc, t_leaf); else hipLaunchKernelGGL((k_trace_ref<float, false, false>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, t_refill, t_desc, t_leaf); }
        break;
    case RT_MODE_FP64:
        if (d.linked) { block = dim3(kTopBlock); grid = dim3((grid.x * 256 + kTopBlock - 1) / kTopBlock); }
        if (d_counters)
            { if (d.linked) hipLaunchKernelGGL((k_trace_ref<double, true, true>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, t_refill, t_desc, t_leaf); else hipLaunchKernelGGL((k_trace_ref<double, true, false>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, t_refill, t_desc, t_leaf); }
        else
            { if (d.linked && guided) hipLaunchKernelGGL((k_trace_ref<double, false, true, true>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, t_refill, t_desc, t_leaf); else if (d.linked) hipLaunchKernelGGL((k_trace_ref<double, false, true>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, t_refill, t_desc, t_leaf); else hipLaunchKernelGGL((k_trace_ref<double, false, false>), grid, block, 0, stream, d, d_rays, d_hits, n32, d_n, cursor, d_counters, t_refill, t_desc, t_leaf); }
        break;
    default:
        return fail(RT_ERR_INVALID, "unknown traversal mode " + std::to_string(mode));
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

rt_status trace_rays_host(Scene &sc, const rt_ray *rays, size_t n, rt_hit *hits, int mode,
                          bool want_counters, rt_stats *stats) {
    DeviceGuard guard(sc.device);
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n == 0) return RT_OK;
    if (!rays || !hits) return fail(RT_ERR_INVALID, "rays/hits is NULL");
    rt_ray *d_rays = nullptr;
    rt_hit *d_hits = nullptr;
    unsigned long long *d_cnt = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    rt_status st = RT_OK;
    auto cleanup = [&]() {
        if (d_rays) (void)hipFree(d_rays);
        if (d_hits) (void)hipFree(d_hits);
        if (d_cnt) (void)hipFree(d_cnt);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define TRY_CLEAN(expr)                                                                   \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) {                                                          \
            cleanup();                                                                    \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
        }                                                                                 \
    } while (0)
    TRY_CLEAN(hipMalloc(reinterpret_cast<void **>(&d_rays), n * sizeof(rt_ray)));
    TRY_CLEAN(hipMalloc(reinterpret_cast<void **>(&d_hits), n * sizeof(rt_hit)));
    TRY_CLEAN(hipMemcpy(d_rays, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice));
    if (want_counters) {
        TRY_CLEAN(hipMalloc(reinterpret_cast<void **>(&d_cnt), 8 * sizeof(unsigned long long)));
        TRY_CLEAN(hipMemset(d_cnt, 0, 8 * sizeof(unsigned long long)));
    }
    TRY_CLEAN(hipEventCreate(&e0));
    TRY_CLEAN(hipEventCreate(&e1));
    TRY_CLEAN(hipEventRecord(e0, nullptr));
    st = launch_trace(sc, d_rays, n, nullptr, d_hits, mode, d_cnt, nullptr);
    if (st != RT_OK) {
        cleanup();
        return st;
    }
    TRY_CLEAN(hipEventRecord(e1, nullptr));
    TRY_CLEAN(hipEventSynchronize(e1));
    TRY_CLEAN(hipMemcpy(hits, d_hits, n * sizeof(rt_hit), hipMemcpyDeviceToHost));
    if (stats) {
        float ms = 0;
        TRY_CLEAN(hipEventElapsedTime(&ms, e0, e1));
        stats->kernel_ms = ms;
        stats->total_ms = ms;
        stats->kernel_launches = 1;
        stats->rays = n;
        if (want_counters) {
            unsigned long long c[5];
            TRY_CLEAN(hipMemcpy(c, d_cnt, sizeof c, hipMemcpyDeviceToHost));
            stats->nodes_visited = c[1];
            stats->box_tests = c[2];
            stats->tris_fetched = c[3];
            stats->tris_tested = c[4];
        }
    }
#undef TRY_CLEAN
    cleanup();
    return RT_OK;
}

// ---------------------------------------------------------------------------
// Path workspace: every per-batch buffer is a range of ONE block of device
// memory, so that the block can come from hipMalloc (grow-only, kept by the
// scene) or from the library's static device array alike.
// ---------------------------------------------------------------------------
struct WsLayout {
    size_t rays[2], hits[2], counts, head, slots, terms_a, terms_b, radiance, ray_total, trav, cursor, path_rays, bins;
    size_t total;
};

static WsLayout ws_layout(size_t P, int L, int B) {
    const size_t Lc = (size_t)(L > 0 ? L : 1), qcap = P * (Lc + 1), nv = (size_t)B + 1;
    WsLayout l{};
    size_t at = 0;
    auto take = [&](size_t bytes) {
        const size_t here = at;
        at += (bytes + 255) & ~(size_t)255;
        return here;
    };
    for (int i = 0; i < 2; ++i) {
        l.rays[i] = take(qcap * sizeof(rt_ray));
        l.hits[i] = take(qcap * sizeof(rt_hit));
    }
    l.counts = take(((size_t)B + 3) * sizeof(uint32_t));
    l.head = take(P * sizeof(PathHead));
    l.slots = take(P * Lc * sizeof(f4));
    l.terms_a = take(P * nv * sizeof(f4));
    l.terms_b = take(P * nv * sizeof(Pair32));
    l.radiance = take(P * 3 * sizeof(float));
    l.ray_total = take(64);
    l.trav = take(8 * sizeof(unsigned long long));
    l.cursor = take((size_t)kCursorStride * sizeof(uint32_t));
    l.path_rays = take(P * sizeof(uint32_t));
    l.bins = take((size_t)kProgressBins * sizeof(unsigned long long));
    l.total = at;
    return l;
}

static PathBuffers ws_carve(void *block, const WsLayout &l) {
    char *base = static_cast<char *>(block);
    PathBuffers b{};
    for (int i = 0; i < 2; ++i) {
        b.rays[i] = reinterpret_cast<rt_ray *>(base + l.rays[i]);
        b.hits[i] = reinterpret_cast<rt_hit *>(base + l.hits[i]);
    }
    b.counts = reinterpret_cast<uint32_t *>(base + l.counts);
    b.head = reinterpret_cast<PathHead *>(base + l.head);
    b.slots = reinterpret_cast<f4 *>(base + l.slots);
    b.terms_a = reinterpret_cast<f4 *>(base + l.terms_a);
    b.terms_b = reinterpret_cast<Pair32 *>(base + l.terms_b);
    b.radiance = reinterpret_cast<float *>(base + l.radiance);
    b.ray_total = reinterpret_cast<unsigned long long *>(base + l.ray_total);
    b.trav_counters = reinterpret_cast<unsigned long long *>(base + l.trav);
    b.cursor = reinterpret_cast<uint32_t *>(base + l.cursor);
    b.path_rays = reinterpret_cast<uint32_t *>(base + l.path_rays);
    b.progress_bins = reinterpret_cast<unsigned long long *>(base + l.bins);
    return b;
}

size_t workspace_bytes_per_path(int L, int B) {
    const size_t Lc = (size_t)(L > 0 ? L : 1), nv = (size_t)(B > 0 ? B : 0) + 1;
    return 2 * (Lc + 1) * (sizeof(rt_ray) + sizeof(rt_hit)) + sizeof(PathHead) + Lc * sizeof(f4) +
           nv * (sizeof(f4) + sizeof(Pair32)) + 3 * sizeof(float) + sizeof(uint32_t);
}

// Grow-only workspace for `paths` simultaneous paths with L light samples and
// B bounces, reused across batches and calls.
static rt_status workspace_ensure(Scene &sc, size_t paths, int L, int B, PathBuffers &w) {
    if (sc.ws_paths < paths || sc.ws_L < L || sc.ws_B < B) {
        workspace_free(sc);
        const WsLayout l = ws_layout(paths, L, B);
        // a block that does not fit (free device memory, or the scene's own bound)
        // is reported as RT_ERR_NOMEM so that the caller can retry with a smaller batch
        if (sc.ws_byte_limit && l.total > sc.ws_byte_limit)
            return fail(RT_ERR_NOMEM, "path workspace of " + std::to_string(l.total) +
                                          " bytes exceeds the scene's workspace limit");
        void *block = nullptr;
        const hipError_t e = hipMalloc(&block, l.total);
        if (e != hipSuccess) {
            (void)hipGetLastError();   // clear the sticky error so that a smaller retry can succeed
            return fail(e == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE,
                        std::string("hipMalloc of the path workspace: ") + hipGetErrorString(e));
        }
        sc.ws_block = block;
        sc.ws_block_bytes = l.total;
        sc.ws = ws_carve(block, l);
        sc.ws.cursor = nullptr;        // allocated workspaces use the scene's cursor ring
        sc.ws_paths = paths;
        sc.ws_L = L;
        sc.ws_B = B;
    }
    w = sc.ws;
    return RT_OK;
}

// Timing events of one call, destroyed on every way out of it.
struct EventPool {
    std::vector<hipEvent_t> all;
    ~EventPool() {
        for (hipEvent_t e : all) (void)hipEventDestroy(e);
    }
    hipError_t make(hipEvent_t *e) {
        const hipError_t r = hipEventCreate(e);
        if (r == hipSuccess) all.push_back(*e);
        return r;
    }
};

// True when two passes differ only in what may differ between tiles of one frame.
static bool same_frame(const rt_pass &a, const rt_pass &b) {
    return std::memcmp(&a.camera, &b.camera, sizeof a.camera) == 0 &&
           std::memcmp(&a.settings, &b.settings, sizeof a.settings) == 0 &&
           a.total_width == b.total_width && a.total_height == b.total_height;
}

// The render loop behind every tile entry point.  `fixed` != nullptr: the path
// workspace is that block (fixed_paths paths at most) instead of allocated
// memory, and the call allocates nothing, neither on the device nor through
// the HIP runtime (one pass, no statistics).
static rt_status render_core(Scene &sc, const rt_pass *passes, int32_t n_passes, int mode,
                             int32_t sample_begin, int32_t sample_count, bool incremental,
                             bool zero_accum, float *d_accum, uint8_t *d_rgba, rt_stats *stats,
                             hipStream_t stream, FixedWorkspace *fixed, size_t fixed_paths,
                             ProgressBins *progress = nullptr) {
    DeviceGuard guard(sc.device);
    if (!guard.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    if (fixed && (stats || n_passes != 1)) return fail(RT_ERR_INVALID, "the static-workspace call takes one pass and no statistics");
    if (progress && (fixed || n_passes != 1 || incremental || progress->n_bins < 1 || progress->n_bins > kProgressBins || progress->step < 1))
        return fail(RT_ERR_INVALID, "progress bins need one whole-tile pass on the scene's own workspace");
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (!passes || n_passes < 1) return fail(RT_ERR_INVALID, "no pass given");
    if (n_passes > 65536) return fail(RT_ERR_INVALID, "more than 65536 tiles in one call");
    if (!d_accum) return fail(RT_ERR_INVALID, "accumulation buffer is NULL");
    if (sample_begin < 0 || sample_count < 0) return fail(RT_ERR_INVALID, "bad sample range");
    if (d_rgba && (int64_t)sample_begin + (int64_t)sample_count < 1)
        return fail(RT_ERR_INVALID, "an image of zero samples cannot be resolved");
    if (mode != RT_MODE_STRICT_F32 && mode != RT_MODE_FAST_F32 && mode != RT_MODE_FP64 && mode != RT_MODE_FAST_NEAR_F32)
        return fail(RT_ERR_INVALID, "unknown traversal mode " + std::to_string(mode));
    const rt_pass &pass = passes[0];
    PassParams pp;
    rt_status st = prepare_pass(sc.light, pass, pp);
    if (st != RT_OK) return st;
    // The ordered traversal takes its near-origin form when every camera ray of the frame
    // starts inside the scene's near region (camera.go:115-121: a ray starts on the
    // projection plane; the frame's four corners bound all of them, the map is affine);
    // shadow and bounce rays start on the scene's surfaces, inside that region by
    // construction.  Tiles of one call share camera and frame.
    if (mode == RT_MODE_FAST_F32 || mode == RT_MODE_FAST_NEAR_F32) {
        bool inside = sc.d_wide != nullptr;
        for (int corner = 0; corner < 4 && inside; ++corner) {
            const float lx = pp.top_left[0] + pp.steps[0] * (float)((corner & 1) ? pass.total_width : 0);
            const float ly = pp.top_left[1] - pp.steps[1] * (float)((corner & 2) ? pass.total_height : 0);
            const v3 o = transform_coordinate(v3{lx, ly, -pp.ppd}, pp.cam_transform);
            const float c[3] = {o.x, o.y, o.z};
            for (int a = 0; a < 3; ++a) inside = inside && c[a] >= sc.near_lo[a] && c[a] <= sc.near_hi[a];
        }
        mode = inside ? RT_MODE_FAST_NEAR_F32 : RT_MODE_FAST_F32;
    }
    pp.incremental = incremental ? 1 : 0;
    size_t pixels = (size_t)pass.width * (size_t)pass.height;
    std::vector<TileDesc> tiles;
    if (n_passes > 1) {
        tiles.resize((size_t)n_passes);
        pixels = 0;
        for (int32_t i = 0; i < n_passes; ++i) {
            const rt_pass &t = passes[i];
            if (!same_frame(pass, t))
                return fail(RT_ERR_INVALID, "tiles of one call must share camera, settings and frame size (tile " + std::to_string(i) + ")");
            if (t.width <= 0 || t.height <= 0)
                return fail(RT_ERR_INVALID, "pass width/height must be positive (tile " + std::to_string(i) + ")");
            tiles[(size_t)i] = TileDesc{t.x_offset, t.y_offset, t.width, t.height, t.rng_seed, (uint32_t)pixels, 0};
            pixels += (size_t)t.width * (size_t)t.height;
        }
    }
    if (pixels > 0xffffffffULL) return fail(RT_ERR_INVALID, "more than 2^32-1 pixels in one call");
    const int L = pp.L, B = pp.B;

    // one render call at a time per workspace, on the host and on the GPU: the
    // scene's own, or the device's static one
    std::mutex &mu = fixed ? fixed->mu : sc.render_mu;
    hipEvent_t done = fixed ? fixed->done : sc.ws_done;
    std::lock_guard<std::mutex> lock(mu);
    // A stream that is being captured into a graph takes no part in the event chain:
    // an event recorded during capture belongs to the graph and could not be waited
    // for by a later, uncaptured call.  Replays of such graphs are ordered by the
    // caller (documented with rt_render_tile_static).
    bool capturing = false;
    {
        hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing(stream, &status) == hipSuccess)
            capturing = status == hipStreamCaptureStatusActive;
        else
            (void)hipGetLastError();
    }
    if (!capturing) HIP_TRY(hipStreamWaitEvent(stream, done, 0));
    struct MarkDone {
        hipEvent_t done;
        hipStream_t stream;
        bool skip;
        ~MarkDone() {
            if (!skip) (void)hipEventRecord(done, stream);
        }
    } mark_done{done, stream, capturing};

    // batch shape: whole pixel rows of the tile-linear range x as many samples as fit
    const size_t Lq = (size_t)(L > 0 ? L : 1) + 1;             // queue slots a path can fill per vertex
    size_t max_paths = fixed ? fixed_paths : sc.max_paths_in_flight;
    // a traversal launch takes at most 2^31-1 rays: the queue of one batch must stay below that
    while (max_paths > 64 && max_paths * Lq > (size_t)0x7fffffff) max_paths /= 2;
    if (max_paths * Lq > (size_t)0x7fffffff)
        return fail(RT_ERR_INVALID, "LightSampleRays too large: a ray queue would exceed 2^31-1 rays");
    size_t n_pix = pixels < max_paths ? pixels : max_paths;
    if (progress && progress->chunks > 1) {
        const size_t range = (pixels + (size_t)progress->chunks - 1) / (size_t)progress->chunks;
        if (range >= 1 && range < n_pix) n_pix = range;
    }
    size_t n_samp = max_paths / n_pix;
    if (n_samp < 1) n_samp = 1;
    if (n_samp > (size_t)sample_count) n_samp = (size_t)(sample_count > 0 ? sample_count : 1);
    PathBuffers w;
    if (fixed) {
        const WsLayout l = ws_layout(n_pix * n_samp, L, B);
        if (l.total > fixed->bytes) return fail(RT_ERR_INVALID, "batch does not fit the given workspace block");
        w = ws_carve(fixed->ptr, l);
    } else {
        // A call that is being captured into a graph must not allocate or free: that
        // would invalidate the capture, and freeing could take the workspace from
        // under a call still running on another stream.
        if (capturing && (sc.ws_paths < n_pix * n_samp || sc.ws_L < L || sc.ws_B < B ||
                          (!tiles.empty() && sc.tile_cap < tiles.size())))
            return fail(RT_ERR_STATE, "the path workspace would have to grow during stream capture: render once "
                                      "with these settings before capturing, or use rt_render_tile_static");
        // the workspace is sized by the batch; when it does not fit (free device
        // memory, or the scene's workspace limit), halve the batch (down to 64 k
        // samples) and remember the smaller bound
        for (;;) {
            st = workspace_ensure(sc, n_pix * n_samp, L, B, w);
            if (st != RT_ERR_NOMEM || n_pix * n_samp <= ((size_t)1 << 16)) break;
            if (n_samp > 1) n_samp = (n_samp + 1) / 2;
            else n_pix = (n_pix + 1) / 2;
            sc.max_paths_in_flight = n_pix * n_samp;
        }
        if (st != RT_OK) return st;
    }
    if (!tiles.empty()) {
        if (sc.tile_cap < tiles.size()) {
            if (sc.d_tiles) (void)hipFree(sc.d_tiles);
            sc.d_tiles = nullptr;
            sc.tile_cap = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc.d_tiles), tiles.size() * sizeof(TileDesc)));
            sc.tile_cap = tiles.size();
        }
        HIP_TRY(hipMemcpyAsync(sc.d_tiles, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));   // `tiles` is pageable host memory
        pp.n_tiles = (int32_t)tiles.size();
        pp.tiles = sc.d_tiles;
    }

    const DevScene d = dev_view(sc);
    if (progress) {
        if ((st = zero_words(w.progress_bins, kProgressBins * sizeof(unsigned long long), stream)) != RT_OK) return st;
    } else {
        w.path_rays = nullptr;             // the kernels keep no per-path ray counts
    }
    if (zero_accum) {
        if ((st = zero_words(d_accum, pixels * 12, stream)) != RT_OK) return st;
    }
    if (stats) {
        if ((st = zero_words(w.ray_total, 8, stream)) != RT_OK) return st;
        if ((st = zero_words(w.trav_counters, 8 * sizeof(unsigned long long), stream)) != RT_OK) return st;
    }
    EventPool events;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_events;
    if (stats) {
        HIP_TRY(events.make(&e0));
        HIP_TRY(events.make(&e1));
        HIP_TRY(hipEventRecord(e0, stream));
    }
    auto timed_trace = [&](const rt_ray *rays, size_t cap, const uint32_t *cnt, rt_hit *hits) -> rt_status {
        hipEvent_t a = nullptr, b = nullptr;
        if (stats) {
            HIP_TRY(events.make(&a));
            HIP_TRY(events.make(&b));
            trace_events.emplace_back(a, b);
            HIP_TRY(hipEventRecord(a, stream));
        }
        rt_status s2 = launch_trace(sc, rays, cap, cnt, hits, mode,
                                    (stats && sc.count_traversal) ? w.trav_counters : nullptr, stream,
                                    w.cursor);
        if (s2 != RT_OK) return s2;
        if (stats) HIP_TRY(hipEventRecord(b, stream));
        return RT_OK;
    };

    for (size_t pix0 = 0; pix0 < pixels; pix0 += n_pix) {
        const size_t np = (pixels - pix0) < n_pix ? (pixels - pix0) : n_pix;
        for (size_t s0 = 0; s0 < (size_t)sample_count; s0 += n_samp) {
            const size_t ns = ((size_t)sample_count - s0) < n_samp ? ((size_t)sample_count - s0) : n_samp;
            Batch b;
            b.P = (uint32_t)(np * ns);
            b.n_pix = (uint32_t)np;
            b.pix0 = (uint32_t)pix0;
            b.s0 = (uint32_t)(sample_begin + (int32_t)s0);
            b.n_samp = (uint32_t)ns;
            const dim3 block(256), grid((b.P + 255) / 256);
            const size_t qcap = (size_t)b.P * (size_t)((L > 0 ? L : 1) + 1);
            if ((st = zero_words(w.counts, (size_t)(B + 3) * sizeof(uint32_t), stream)) != RT_OK) return st;
            hipLaunchKernelGGL(k_raygen, grid, block, 0, stream, d, pp, b, w);
            if ((st = timed_trace(w.rays[0], b.P, w.counts + 0, w.hits[0])) != RT_OK) return st;
            for (int k = 0; k <= B; ++k) {
                hipLaunchKernelGGL(k_shade, dim3((b.P + kShadeBlock - 1) / kShadeBlock), dim3(kShadeBlock), 0, stream, d, pp, b, w, k);
                const int o = (k + 1) & 1;
                if ((st = timed_trace(w.rays[o], qcap, w.counts + k + 1, w.hits[o])) != RT_OK) return st;
            }
            hipLaunchKernelGGL(k_finish, grid, block, 0, stream, d, pp, b, w, (B + 1) & 1);
            hipLaunchKernelGGL(k_accumulate, dim3((b.n_pix + 255) / 256), block, 0, stream, b, w.radiance, d_accum);
            if (stats) hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(64), 0, stream, w.counts, B + 2, w.ray_total);
            if (progress && w.path_rays)
                hipLaunchKernelGGL(k_progress_bins, grid, block, 0, stream, pp, b, w.path_rays, progress->step,
                                   progress->n_bins, w.progress_bins);
            HIP_TRY(hipGetLastError());
        }
        // a pixel range is done: report the progress bins that are complete by now (the camera
        // samples of pixel i have the indices i * spp .. i * spp + spp - 1, main.go:132-152)
        if (progress && progress->on_range && pix0 + np < pixels) {
            HIP_TRY(hipMemcpyAsync(progress->rays, w.progress_bins, kProgressBins * sizeof(unsigned long long),
                                   hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            const unsigned long long done = (unsigned long long)(pix0 + np) * (unsigned long long)sample_count;
            unsigned long long complete = done / progress->step;
            if (complete > (unsigned long long)progress->n_bins) complete = (unsigned long long)progress->n_bins;
            progress->on_range((int32_t)complete);
        }
    }
    if (d_rgba) {
        const int32_t divisor = sample_begin + sample_count;
        const float inv = 1.0f / (float)divisor;
        const double inv_gamma = (double)(1.0f / pass.settings.gamma);
        hipLaunchKernelGGL(k_resolve, dim3((uint32_t)((pixels + 255) / 256)), dim3(256), 0, stream,
                           d_accum, pixels, inv, pass.settings.gamma_correction, inv_gamma, d_rgba);
        HIP_TRY(hipGetLastError());
    }
    if (progress) {
        HIP_TRY(hipMemcpyAsync(progress->rays, w.progress_bins, kProgressBins * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    if (stats) {
        HIP_TRY(hipEventRecord(e1, stream));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        stats->total_ms = ms;
        double kms = 0;
        for (auto &pr : trace_events) {
            float t = 0;
            HIP_TRY(hipEventElapsedTime(&t, pr.first, pr.second));
            kms += t;
        }
        stats->kernel_ms = kms;
        stats->kernel_launches = trace_events.size();
        if (std::getenv("RT_DIAG") && trace_events.size() >= (size_t)(B + 2)) {
            // the last batch: queue lengths and launch times per path depth
            std::vector<uint32_t> counts((size_t)(B + 2));
            HIP_TRY(hipMemcpy(counts.data(), w.counts, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (int k = 0; k < B + 2; ++k) {
                float t = 0;
                const auto &pr = trace_events[trace_events.size() - (size_t)(B + 2) + (size_t)k];
                HIP_TRY(hipEventElapsedTime(&t, pr.first, pr.second));
                std::fprintf(stderr, "[rt diag] last batch, queue %d: %u rays in %.3f ms = %.0f Mrays/s\n", k,
                             counts[(size_t)k], t, t > 0 ? counts[(size_t)k] / t / 1e3 : 0.0);
            }
        }
        unsigned long long total = 0;
        HIP_TRY(hipMemcpy(&total, w.ray_total, 8, hipMemcpyDeviceToHost));
        stats->rays = total;
        if (sc.count_traversal) {
            unsigned long long c[8];
            HIP_TRY(hipMemcpy(c, w.trav_counters, sizeof c, hipMemcpyDeviceToHost));
            if (std::getenv("RT_DIAG"))
                std::fprintf(stderr, "[rt diag] lane-visits A %llu wave-iters A %llu (util %.3f) lane-iters B %llu wave-iters B %llu (util %.3f)\n",
                             c[1], c[5], c[5] ? (double)c[1] / (64.0 * c[5]) : 0.0, c[7], c[6], c[6] ? (double)c[7] / (64.0 * c[6]) : 0.0);
            stats->nodes_visited = c[1];
            stats->box_tests = c[2];
            stats->tris_fetched = c[3];
            stats->tris_tested = c[4];
        }
    }
    return RT_OK;
}

rt_status render_accumulate(Scene &sc, const rt_pass *passes, int32_t n_passes, int mode,
                            int32_t sample_begin, int32_t sample_count, bool incremental,
                            bool zero_accum, float *d_accum, uint8_t *d_rgba, rt_stats *stats,
                            hipStream_t stream) {
    return render_core(sc, passes, n_passes, mode, sample_begin, sample_count, incremental, zero_accum,
                       d_accum, d_rgba, stats, stream, nullptr, 0);
}

rt_status render_tile_static(Scene &sc, const rt_pass &pass, int mode, float *d_accum, uint8_t *d_rgba,
                             hipStream_t stream) {
    if (sc.device < 0 || sc.device >= kMaxDevices || !g_static[sc.device].ready)
        return fail(RT_ERR_STATE, "the static workspace of this device was not set up");
    if (!d_rgba) return fail(RT_ERR_INVALID, "rgba is NULL");
    if (pass.width <= 0 || pass.height <= 0) return fail(RT_ERR_INVALID, "pass width/height must be positive");
    const int L = pass.settings.light_sample_rays, B = pass.settings.bounce_limit & 0xff;
    if (L < 0) return fail(RT_ERR_INVALID, "LightSampleRays is negative");
    // the largest batch whose buffers fit the array (every buffer starts on a 256-byte boundary)
    const size_t per_path = workspace_bytes_per_path(L, B);
    size_t paths = kStaticScratchBytes / per_path;
    while (paths > 0 && ws_layout(paths, L, B).total > kStaticScratchBytes) paths -= paths > 64 ? 64 : 1;
    const size_t pixels = (size_t)pass.width * (size_t)pass.height;
    if (pixels > paths)
        return fail(RT_ERR_INVALID,
                    "tile too large for the static workspace: " + std::to_string(pixels) + " pixels x " +
                        std::to_string(per_path) + " bytes per path at LightSampleRays " + std::to_string(L) +
                        ", BounceLimit " + std::to_string(B) + " exceed its " +
                        std::to_string(kStaticScratchBytes) + " bytes (at most " + std::to_string(paths) +
                        " pixels per call at these settings)");
    return render_core(sc, &pass, 1, mode, 0, pass.camera.rays_per_pixel, false, true, d_accum, d_rgba,
                       nullptr, stream, &g_static[sc.device], paths);
}

// ---------------------------------------------------------------------------
// Gather probe: the access pattern of tree traversal in isolation.  Every lane
// chases random 64-byte records (four 16-byte loads; the next index depends on
// the data just read) through a table of a given size, at full occupancy.  The
// rate is the ceiling the memory system sets for a traversal whose records live
// in a table of that size: L2 / Infinity Cache resident below 256 MB, HBM
// beyond.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_gather_probe(const uint4 *__restrict__ rec, uint32_t n_rec, int steps, uint32_t *out) {
    uint32_t state = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    uint32_t idx = (uint32_t)(((unsigned long long)state * n_rec) >> 32);
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        const uint4 *p = rec + 4 * (size_t)idx;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        const uint32_t v = a.x ^ b.y ^ c.z ^ d.w;
        acc += v;
        // a full 32-bit generator state per lane: a walk on the indices themselves
        // would fall into short cycles shared by all lanes and hit in cache
        state = state * 1664525u + 1013904223u + (v & 1u);
        idx = (uint32_t)(((unsigned long long)state * n_rec) >> 32);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void k_fill_probe(uint32_t *words, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) words[i] = (uint32_t)(i * 2654435761u) >> 7;
}

rt_status gather_probe(int device, size_t table_bytes, double *records_per_s) {
    if (!records_per_s) return fail(RT_ERR_INVALID, "NULL argument");
    if (table_bytes < 64 || table_bytes > ((size_t)1 << 36)) return fail(RT_ERR_INVALID, "table size out of range");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const uint32_t n_rec = (uint32_t)std::min<size_t>(table_bytes / 64, 0xffffffffu);
    const int blocks = prop.multiProcessorCount * 8, steps = 2000;
    uint4 *d_rec = nullptr;
    uint32_t *d_out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    rt_status st = RT_OK;
    auto cleanup = [&]() {
        if (d_rec) (void)hipFree(d_rec);
        if (d_out) (void)hipFree(d_out);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define PROBE_TRY(expr)                                                                  \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            cleanup();                                                                   \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
        }                                                                                \
    } while (0)
    PROBE_TRY(hipMalloc(reinterpret_cast<void **>(&d_rec), (size_t)n_rec * 64));
    PROBE_TRY(hipMalloc(reinterpret_cast<void **>(&d_out), (size_t)blocks * 256 * 4));
    PROBE_TRY(hipEventCreate(&e0));
    PROBE_TRY(hipEventCreate(&e1));
    const size_t words = (size_t)n_rec * 16;
    hipLaunchKernelGGL(k_fill_probe, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, nullptr,
                       reinterpret_cast<uint32_t *>(d_rec), words);
    float best = 0;
    for (int rep = 0; rep < 3; ++rep) {          // the first run warms the caches
        PROBE_TRY(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(k_gather_probe, dim3(blocks), dim3(256), 0, nullptr, d_rec, n_rec, steps, d_out);
        PROBE_TRY(hipEventRecord(e1, nullptr));
        PROBE_TRY(hipEventSynchronize(e1));
        float ms = 0;
        PROBE_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && (best == 0 || ms < best)) best = ms;
    }
#undef PROBE_TRY
    cleanup();
    if (!(best > 0)) return fail(RT_ERR_DEVICE, "gather probe measured no time");
    *records_per_s = (double)blocks * 256.0 * steps / (best * 1e-3);
    return st;
}

rt_status resolve_device(const float *d_accum, size_t pixels, int32_t divisor, int gamma_on,
                         float gamma, uint8_t *d_rgba, hipStream_t stream) {
    if (!d_accum || !d_rgba) return fail(RT_ERR_INVALID, "resolve buffers are NULL");
    if (divisor < 1) return fail(RT_ERR_INVALID, "divisor must be positive");
    if (pixels == 0) return RT_OK;
    const float inv = 1.0f / (float)divisor;
    const double inv_gamma = (double)(1.0f / gamma);
    hipLaunchKernelGGL(k_resolve, dim3((uint32_t)((pixels + 255) / 256)), dim3(256), 0, stream,
                       d_accum, pixels, inv, gamma_on, inv_gamma, d_rgba);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

rt_status render_tile_host(Scene &sc, const rt_pass &pass, int mode, float *accum_rgb,
                           uint8_t *rgba, rt_stats *stats, ProgressBins *progress) {
    DeviceGuard guard(sc.device);
    if (!rgba) return fail(RT_ERR_INVALID, "rgba is NULL");
    if (pass.width <= 0 || pass.height <= 0) return fail(RT_ERR_INVALID, "pass width/height must be positive");
    const size_t pixels = (size_t)pass.width * (size_t)pass.height;
    float *d_accum = nullptr;
    uint8_t *d_rgba = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_accum), pixels * 12));
    if (hipMalloc(reinterpret_cast<void **>(&d_rgba), pixels * 4) != hipSuccess) {
        (void)hipFree(d_accum);
        return fail(RT_ERR_NOMEM, "hipMalloc of the tile image failed");
    }
    rt_stats local;
    rt_status st = render_core(sc, &pass, 1, mode, 0, pass.camera.rays_per_pixel, false, true, d_accum,
                               d_rgba, stats ? stats : &local, nullptr, nullptr, 0, progress);
    if (st == RT_OK) {
        hipError_t e = hipMemcpy(rgba, d_rgba, pixels * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && accum_rgb) e = hipMemcpy(accum_rgb, d_accum, pixels * 12, hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = fail(RT_ERR_DEVICE, std::string("copy-back failed: ") + hipGetErrorString(e));
    }
    (void)hipFree(d_accum);
    (void)hipFree(d_rgba);
    return st;
}

}  // namespace rt
