// Micro-benchmark: how fast can a CU fetch one random 64-byte record per lane?
//   variant 0: every lane reads its own record with four 16-byte loads (divergent)
//   variant 1: the four lanes of a quad read one lane's record together, one
//              16-byte piece each (coalesced within the quad), four turns
//   variant 2: variant 1 + exchange through LDS so that each lane ends up with
//              its own 64 bytes
// The next record index depends on the data just read (pointer chasing), as in
// tree traversal.  Build: hipcc --offload-arch=gfx950 -O3 gather.hip -o gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct f4 { float x, y, z, w; };
typedef unsigned int u32;

template <int VARIANT>
__global__ void __launch_bounds__(256) k(const uint4 *__restrict__ rec, u32 n_rec, int steps, u32 *out) {
    __shared__ uint4 xch[4 * 4 * 64];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u % n_rec;
    u32 acc = 0;
    for (int s = 0; s < steps; ++s) {
        uint4 a, b, c, d;
        if (VARIANT == 0) {
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
        if (VARIANT == 1) {
            // every lane saw a different mix; the own record's words are spread over the quad
            idx = (idx * 1664525u + 1013904223u + (v & 1u)) % n_rec;
        } else {
            idx = (idx * 1664525u + 1013904223u + (v & 1u)) % n_rec;
        }
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
    hipMalloc(&d_rec, h.size() * 4);
    hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    hipMemcpy(d_rec, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 3; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d_rec, n_rec, steps, d_out);
            if (v == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d_rec, n_rec, steps, d_out);
            if (v == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d_rec, n_rec, steps, d_out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double recs = (double)blocks * 256 * steps;
            if (rep) printf("variant %d: %.2f ms, %.1f G records/s, %.0f GB/s\n", v, ms, recs / ms / 1e6, recs * 64 / ms / 1e6);
        }
    }
    return 0;
}
