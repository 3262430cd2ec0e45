// gomath.h -- float64 helpers with the semantics of Go's pure-Go math package
// (the implementations the reference runs under js/wasm), usable from host
// and device code.  The shading kernels call them where the reference calls
// math.Cos (process/trace.go:103-104), math.Max / math.Min (trace.go:65-66,
// utility/utility.go:34-40) and math.Pow (main.go:164-171).  Compile with
// -ffp-contract=off: every operation rounds on its own.
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ inline
#else
#define RT_HD inline
#endif

namespace gomath {

constexpr double kPi = 3.14159265358979323846264338327950288419716939937510582097494459;
constexpr double kSqrt2 = 1.41421356237309504880168872420969807856967187537694807317667974;

RT_HD bool is_nan(double x) { return x != x; }
RT_HD bool is_inf(double x) { return x == INFINITY || x == -INFINITY; }
RT_HD bool sign_bit(double x) {
    union { double d; uint64_t u; } c;
    c.d = x;
    return (c.u >> 63) != 0;
}

// math.Max: +Inf wins, NaN propagates, Max(+0,-0) = +0
RT_HD double go_max(double x, double y) {
    if (x == INFINITY || y == INFINITY) return INFINITY;
    if (is_nan(x) || is_nan(y)) return NAN;
    if (x == 0 && x == y) return sign_bit(x) ? y : x;
    return x > y ? x : y;
}
// math.Min: -Inf wins, NaN propagates, Min(-0,+0) = -0
RT_HD double go_min(double x, double y) {
    if (x == -INFINITY || y == -INFINITY) return -INFINITY;
    if (is_nan(x) || is_nan(y)) return NAN;
    if (x == 0 && x == y) return sign_bit(x) ? x : y;
    return x < y ? x : y;
}

// math.Sin / math.Cos (math/sin.go, Cephes-derived).  Cody-Waite reduction
// only: callers pass bounded arguments (dot products, half-angles), far below
// the 2^29 threshold where Go switches to Payne-Hanek.
RT_HD double sincos_kernel(double x, bool want_cos) {
    const double PI4A = 7.85398125648498535156e-1;
    const double PI4B = 3.77489470793079817668e-8;
    const double PI4C = 2.69515142907905952645e-15;
    if (is_nan(x) || is_inf(x)) return NAN;
    bool sign = false;
    if (x < 0) {
        x = -x;
        if (!want_cos) sign = true;
    }
    uint64_t j = (uint64_t)(x * (4.0 / kPi));
    double y = (double)j;
    if (j & 1) {
        j++;
        y++;
    }
    j &= 7;
    double z = ((x - y * PI4A) - y * PI4B) - y * PI4C;
    if (j > 3) {
        j -= 4;
        sign = !sign;
    }
    if (want_cos && j > 1) sign = !sign;
    double zz = z * z;
    bool mid = (j == 1 || j == 2);
    bool cos_poly = want_cos ? !mid : mid;
    if (cos_poly) {
        y = 1.0 - 0.5 * zz +
            zz * zz * ((((((-1.13585365213876817300e-11 * zz) + 2.08757008419747316778e-9) * zz + -2.75573141792967388112e-7) * zz + 2.48015872888517045348e-5) * zz + -1.38888888888730564116e-3) * zz + 4.16666666666665929218e-2);
    } else {
        y = z + z * zz * ((((((1.58962301576546568060e-10 * zz) + -2.50507477628578072866e-8) * zz + 2.75573136213857245213e-6) * zz + -1.98412698295895385996e-4) * zz + 8.33333333332211858878e-3) * zz + -1.66666666666666307295e-1);
    }
    return sign ? -y : y;
}
RT_HD double go_cos(double x) { return sincos_kernel(x, true); }
RT_HD double go_sin(double x) { return sincos_kernel(x, false); }

constexpr double kLn2Hi = 6.93147180369123816490e-01;
constexpr double kLn2Lo = 1.90821492927058770002e-10;
constexpr double kLog2e = 1.44269504088896338700e+00;

// math.Exp (math/exp.go)
RT_HD double go_exp(double x) {
    if (is_nan(x) || x == INFINITY) return x;
    if (x == -INFINITY) return 0;
    if (x > 7.09782712893383973096e+02) return INFINITY;
    if (x < -7.45133219101941108420e+02) return 0;
    const double near_zero = 1.0 / (double)(1 << 28);
    if (-near_zero < x && x < near_zero) return 1 + x;
    int k = 0;
    if (x < 0)
        k = (int)(kLog2e * x - 0.5);
    else if (x > 0)
        k = (int)(kLog2e * x + 0.5);
    double hi = x - (double)k * kLn2Hi;
    double lo = (double)k * kLn2Lo;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (1.66666666666666657415e-01 + t * (-2.77777777770155933842e-03 + t * (6.61375632143793436117e-05 + t * (-1.65339022054652515390e-06 + t * 4.13813679705723846039e-08))));
    double y = 1 - ((lo - (r * c) / (2 - c)) - hi);
    return ldexp(y, k);
}

// math.Log (math/log.go)
RT_HD double go_log(double x) {
    if (is_nan(x) || x == INFINITY) return x;
    if (x < 0) return NAN;
    if (x == 0) return -INFINITY;
    int ki;
    double f1 = frexp(x, &ki);
    if (f1 < kSqrt2 / 2) {
        f1 *= 2;
        ki--;
    }
    double f = f1 - 1;
    double k = (double)ki;
    double s = f / (2 + f);
    double s2 = s * s;
    double s4 = s2 * s2;
    double t1 = s2 * (6.666666666666735130e-01 + s4 * (2.857142874366239149e-01 + s4 * (1.818357216161805012e-01 + s4 * 1.479819860511658591e-01)));
    double t2 = s4 * (3.999999999940941908e-01 + s4 * (2.222219843214978396e-01 + s4 * 1.531383769920937332e-01));
    double R = t1 + t2;
    double hfsq = 0.5 * f * f;
    return k * kLn2Hi - ((hfsq - (s * (hfsq + R) + k * kLn2Lo)) - f);
}

RT_HD bool is_odd_int(double x) {
    double ip;
    double fp = modf(x, &ip);
    return fp == 0 && (((int64_t)ip) & 1) == 1;
}

// math.Pow (math/pow.go)
RT_HD double go_pow(double x, double y) {
    if (y == 0 || x == 1) return 1;
    if (y == 1) return x;
    if (is_nan(x) || is_nan(y)) return NAN;
    if (x == 0) {
        if (y < 0) {
            if (is_odd_int(y)) return sign_bit(x) ? -INFINITY : INFINITY;
            return INFINITY;
        }
        if (is_odd_int(y)) return x;
        return 0;
    }
    if (is_inf(y)) {
        if (x == -1) return 1;
        if ((fabs(x) < 1) == (y > 0)) return 0;
        return INFINITY;
    }
    if (is_inf(x)) {
        if (x < 0) {
            if (y < 0) return is_odd_int(y) ? -0.0 : 0.0;
            return is_odd_int(y) ? -INFINITY : INFINITY;
        }
        return y < 0 ? 0 : INFINITY;
    }
    if (y == 0.5) return sqrt(x);
    if (y == -0.5) return 1 / sqrt(x);
    double yi;
    double yf = modf(fabs(y), &yi);
    if (yf != 0 && x < 0) return NAN;
    if (yi >= 9.223372036854775808e18) {
        if (x == -1) return 1;
        if ((fabs(x) < 1) == (y > 0)) return 0;
        return INFINITY;
    }
    double a1 = 1.0;
    int ae = 0;
    if (yf != 0) {
        if (yf > 0.5) {
            yf--;
            yi++;
        }
        a1 = go_exp(yf * go_log(x));
    }
    int xe;
    double x1 = frexp(x, &xe);
    for (int64_t i = (int64_t)yi; i != 0; i >>= 1) {
        if (xe < -(1 << 12) || (1 << 12) < xe) {
            ae += xe;
            break;
        }
        if (i & 1) {
            a1 *= x1;
            ae += xe;
        }
        x1 *= x1;
        xe <<= 1;
        if (x1 < .5) {
            x1 += x1;
            xe--;
        }
    }
    if (y < 0) {
        a1 = 1 / a1;
        ae = -ae;
    }
    return ldexp(a1, ae);
}

// utility.ClampColor component (utility/utility.go:34-40)
RT_HD float clamp01(float c) { return (float)go_min(go_max((double)c, 0.0), 1.0); }

}  // namespace gomath
