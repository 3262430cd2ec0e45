// Micro-benchmark: how fast can a CU fetch one random 64-byte record per lane?
//   variant 0: every lane reads its own record with four 16-byte loads (divergent)
//   variant 1: the four lanes of a quad read one lane's record together, one
//              16-byte piece each (coalesced within the quad), four turns
//   variant 2: variant 1 + exchange through LDS so that each lane ends up with
//              its own 64 bytes
//   variants 3, 4, 5: 32 bytes (two loads), 16 bytes (one load), four 8-byte
//              loads per record -- is the cost per load instruction or per byte?
//   variant 6: 128-byte records (eight loads) out of the same table
// The next record index depends on the data just read (pointer chasing), as in
// tree traversal.  Build: hipcc --offload-arch=gfx950 -O3 gather.hip -o gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct f4 { float x, y, z, w; };
typedef unsigned int u32;

template <int VARIANT>
__global__ void __launch_bounds__(256) k(const uint4 *__restrict__ rec, u32 n_rec, int steps, u32 *out) {
    __shared__ uint4 xch[4 * 4 * 64];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 state = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    u32 idx = (u32)(((unsigned long long)state * n_rec) >> 32);
    u32 acc = 0;
    for (int s = 0; s < steps; ++s) {
        uint4 a, b, c, d;
        if (VARIANT == 6) {          // 128-byte records: eight 16-byte loads
            const uint4 *p = rec + 8 * (size_t)(idx >> 1);
            const uint4 e = p[0], f = p[1], g = p[2], hh = p[3], e2 = p[4], f2 = p[5], g2 = p[6], h2 = p[7];
            a = make_uint4(e.x ^ e2.x, e.y, e.z, e.w); b = make_uint4(f.x, f.y ^ f2.y, f.z, f.w);
            c = make_uint4(g.x, g.y, g.z ^ g2.z, g.w); d = make_uint4(hh.x, hh.y, hh.z, hh.w ^ h2.w);
        } else if (VARIANT == 3) {          // half of the record: two 16-byte loads
            const uint4 *p = rec + 4 * (size_t)idx;
            a = p[0]; b = p[1]; c = a; d = b;
        } else if (VARIANT == 4) {   // a quarter: one 16-byte load
            const uint4 *p = rec + 4 * (size_t)idx;
            a = p[0]; b = a; c = a; d = a;
        } else if (VARIANT == 5) {   // four 8-byte loads spread over the record
            const uint2 *p = reinterpret_cast<const uint2 *>(rec + 4 * (size_t)idx);
            const uint2 e = p[0], f = p[2], g = p[4], hh = p[6];
            a = make_uint4(e.x, e.y, 0, 0); b = make_uint4(0, f.x, f.y, 0); c = make_uint4(0, 0, g.x, g.y); d = make_uint4(hh.x, 0, 0, hh.y);
        } else if (VARIANT == 0) {
            const uint4 *p = rec + 4 * (size_t)idx;
            a = p[0]; b = p[1]; c = p[2]; d = p[3];
        } else {
            const u32 q = lane & ~3u, piece = lane & 3u;
            const u32 i0 = __shfl(idx, q + 0), i1 = __shfl(idx, q + 1), i2 = __shfl(idx, q + 2), i3 = __shfl(idx, q + 3);
            a = rec[4 * (size_t)i0 + piece];
            b = rec[4 * (size_t)i1 + piece];
            c = rec[4 * (size_t)i2 + piece];
            d = rec[4 * (size_t)i3 + piece];
            if (VARIANT == 2) {
                uint4 *w = xch + wave * 256;
                // turn j, lane i holds piece (i%4) of lane (q+j): store at [q+j][piece]
                w[(q + 0) * 4 + piece] = a;
                w[(q + 1) * 4 + piece] = b;
                w[(q + 2) * 4 + piece] = c;
                w[(q + 3) * 4 + piece] = d;
                __builtin_amdgcn_wave_barrier();
                a = w[lane * 4 + 0]; b = w[lane * 4 + 1]; c = w[lane * 4 + 2]; d = w[lane * 4 + 3];
                __builtin_amdgcn_wave_barrier();
            }
        }
        const u32 v = a.x ^ b.y ^ c.z ^ d.w;
        acc += v;
        // full 32-bit generator state per lane (a walk on the indices themselves
        // would fall into short cycles shared by all lanes and hit in cache)
        state = state * 1664525u + 1013904223u + (v & 1u);
        idx = (u32)(((unsigned long long)state * n_rec) >> 32);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const u32 n_rec = argc > 1 ? (u32)atoi(argv[1]) : 162487u;
    const int steps = argc > 2 ? atoi(argv[2]) : 2000;
    const int blocks = 256 * (argc > 3 ? atoi(argv[3]) : 6);
    std::vector<u32> h((size_t)n_rec * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (u32)(i * 2654435761u) >> 7;
    uint4 *d_rec; u32 *d_out;
    CHECK(hipMalloc(&d_rec, h.size() * 4));
    CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    CHECK(hipMemcpy(d_rec, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int only = argc > 4 ? atoi(argv[4]) : -1;   // run one variant only (for counter passes)
    for (int v = 0; v < 7; ++v) {
        if (only >= 0 && v != only) continue;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d_rec, n_rec, steps, d_out);
            if (v == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d_rec, n_rec, steps, d_out);
            if (v == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d_rec, n_rec, steps, d_out);
            if (v == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d_rec, n_rec, steps, d_out);
            if (v == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d_rec, n_rec, steps, d_out);
            if (v == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, d_rec, n_rec, steps, d_out);
            if (v == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, d_rec, n_rec, steps, d_out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double recs = (double)blocks * 256 * steps;
            if (rep) printf("variant %d: %.2f ms, %.1f G records/s, %.0f GB/s\n", v, ms, recs / ms / 1e6, recs * 64 / ms / 1e6);
        }
    }
    return 0;
}
