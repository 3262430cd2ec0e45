// rcp_check.hip -- exhaustive check of a short correctly-rounded float32
// reciprocal on the GPU it runs on.
//
// The triangle test divides once per triangle (1.0 / det, triangle.go:86) and a
// ray's set-up three times (1 / direction, ray.go:33-37), and the bit-exact modes
// need the IEEE quotient.  The compiler's expansion of 1.0f / x takes ten
// instructions because it also handles denormal and huge operands by scaling.
// A determinant that passes the cull (det >= 1e-4) and a direction component of
// ordinary size are normal numbers whose reciprocals are normal too, and for
// those five instructions suffice:
//   r0 = v_rcp_f32(x)              (1 ulp)
//   r1 = fma(fma(-x, r0, 1), r0, r0)
//   q  = fma(fma(-x, r1, 1), r1, r1)
// Markstein's theorem promises the correctly rounded quotient when r1 is within
// an ulp; this program does not rely on it and compares q with 1.0f / x for EVERY
// float whose magnitude lies in [2^-100, 2^100], both signs, instead (the range
// rt_core.h's recip3 and recip_det use the short form in; outside it they divide).
// Prints the number of mismatches (expected 0).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt rcp_check.hip -o rcp_check
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ float rcp_short(float x) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
    return __builtin_fmaf(__builtin_fmaf(-x, r1, 1.0f), r1, r1);
}

__global__ void k_check(uint32_t first, uint32_t count, unsigned long long *bad, uint32_t *first_bad) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint32_t bits = first + i;
        const float x = __uint_as_float(bits);
        const float want = 1.0f / x;
        const float got = rcp_short(x);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            if (atomicAdd(bad, 1ull) == 0) *first_bad = bits;
        }
    }
}

int main() {
    const float lo = 0x1p-100f, hi = 0x1p100f;
    uint32_t blo, bhi;
    std::memcpy(&blo, &lo, 4);
    std::memcpy(&bhi, &hi, 4);
    unsigned long long *d_bad = nullptr, bad = 0;
    uint32_t *d_first = nullptr, first = 0;
    if (hipMalloc(&d_bad, 8) != hipSuccess || hipMalloc(&d_first, 4) != hipSuccess) return 2;
    hipMemset(d_bad, 0, 8);
    hipMemset(d_first, 0, 4);
    const uint32_t count = bhi - blo + 1;
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, nullptr, blo, count, d_bad, d_first);                // positive
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, nullptr, blo | 0x80000000u, count, d_bad, d_first);  // negative
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    std::printf("{\"floats_checked\": %llu, \"range\": [%g, %g], \"signs\": \"both\", \"mismatches\": %llu, \"first_bad_bits\": \"0x%08x\"}\n",
                2ull * count, lo, hi, bad, first);
    return bad == 0 ? 0 : 1;
}
