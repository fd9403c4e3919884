// navmath.hpp -- deterministic elementary functions for the gfx950 kernels.
//
// DESIGN.md section 4 specifies these functions operation by operation so that device results
// are bit-identical to the CPU oracle's independent implementation (oracle/navmath_ref.h):
// Cody-Waite pi/2 reduction with two 33-bit pieces + tail, the classic fdlibm minimax
// polynomials for sin/cos/atan/exp, IEEE-754 double add/mul/div only.  Nothing here may be
// contracted into an FMA: the translation unit is compiled with -ffp-contract=off and the
// functions additionally carry the clang pragma.
//
// Why not ocml's sin/cos: the reference's beam direction is cosf/sinf of a float32 heading
// (range_libc, called from env.py:425) and one ulp there can move a sphere-traced hit by a
// whole cell; vendor libm results are not reproducible across host and device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nv {

#pragma clang fp contract(off)

constexpr double kPi = 3.14159265358979311600e+00;
constexpr double kTwoPi = 6.28318530717958623200e+00;

__device__ __forceinline__ double from_bits(uint64_t u) { return __longlong_as_double((long long)u); }

// x = n*(pi/2) + (y0 + y1); |x| < 1e5
__device__ __forceinline__ int rem_pio2(double x, double& y0, double& y1) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00;
    const double pio2_2 = 6.07710050630396597660e-11;
    const double pio2_2t = 2.02226624879595063154e-21;
    double fn = __builtin_rint(x * invpio2);
    double r = x - fn * pio2_1;
    double r2 = r - fn * pio2_2;
    double w = fn * pio2_2t;
    double a = r2 - w;
    y0 = a;
    y1 = (r2 - a) - w;
    return (int)((long long)fn & 3);
}

__device__ __forceinline__ double ksin(double x, double y) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

__device__ __forceinline__ double kcos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double hz = 0.5 * z;
    double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

__device__ __forceinline__ void sincos(double x, double& s, double& c) {
    double y0, y1;
    int n = rem_pio2(x, y0, y1);
    double sn = ksin(y0, y1);
    double cs = kcos(y0, y1);
    double ss = (n & 1) ? cs : sn;
    double cc = (n & 1) ? sn : cs;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}

__device__ __forceinline__ double atan_(double x) {
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
    bool neg = x < 0.0;
    double ax = __builtin_fabs(x);
    double hi = 0.0, lo = 0.0;
    int id;
    if (ax >= 7.3786976294838206464e19) {
        double z = 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
        return neg ? -z : z;
    }
    if (ax < 0.4375) {
        if (ax < 1.862645149230957e-09) return x;
        id = -1;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) { id = 0; hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17;
                           ax = (2.0 * ax - 1.0) / (2.0 + ax); }
        else             { id = 1; hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17;
                           ax = (ax - 1.0) / (ax + 1.0); }
    } else {
        if (ax < 2.4375) { id = 2; hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17;
                           ax = (ax - 1.5) / (1.0 + 1.5 * ax); }
        else             { id = 3; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17;
                           ax = -1.0 / ax; }
    }
    double z = ax * ax;
    double w = z * z;
    double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    double r;
    if (id < 0) r = ax - ax * (s1 + s2);
    else        r = hi - ((ax * (s1 + s2) - lo) - ax);
    return neg ? -r : r;
}

__device__ __forceinline__ double atan2_(double y, double x) {
    const double pi_lo = 1.2246467991473531772e-16;
    const double pi_o_2 = 1.5707963267948965580e+00;
    if (x == 0.0 && y == 0.0) return 0.0;
    if (x == 0.0) return (y < 0.0) ? -pi_o_2 : pi_o_2;
    if (y == 0.0) return (x < 0.0) ? kPi : 0.0;
    double z = atan_(__builtin_fabs(y / x));
    if (x > 0.0) return (y < 0.0) ? -z : z;
    return (y < 0.0) ? (z - pi_lo) - kPi : kPi - (z - pi_lo);
}

// exp(x) for x <= 0 (clamped); x < -700 -> 0
__device__ __forceinline__ double exp_neg(double x) {
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    if (x > 0.0) x = 0.0;
    if (x < -700.0) return 0.0;
    double fk = __builtin_rint(x * invln2);
    double hi = x - fk * ln2HI;
    double lo = fk * ln2LO;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    int k = (int)fk;
    return y * from_bits((uint64_t)(k + 1023) << 52);
}

// Python float % (2*pi)
__device__ __forceinline__ double mod_2pi(double x) {
    // fmod is exact in IEEE arithmetic and returns x itself for |x| < 2*pi -- the usual case (a heading in
    // [0, 2*pi) plus one step of rotation), which then skips the library's iterative reduction
    double m = (__builtin_fabs(x) < kTwoPi) ? x : fmod(x, kTwoPi);
    if (m != 0.0 && m < 0.0) m += kTwoPi;
    return m;
}

// utils.py:5-9 angle_correction
__device__ __forceinline__ double wrap_pi(double a) {
    double s, c;
    sincos(a, s, c);
    return atan2_(s, c);
}

// sqrtf of a non-negative INTEGER below 2^22 held in a float, correctly rounded, in five full-rate-or-
// transcendental instructions: y = v_rsq_f32(x) (1 ulp), s = x * y, one residual step s + (x - s*s) * y / 2 with
// the residual taken exactly by an FMA.  Not correctly rounded in general; on gfx950 it IS for every integer in
// [1, 2^22) -- checked exhaustively against IEEE sqrt on the device (tests/test_gpu_parity.py, all 4 194 303
// inputs; profiles/_diag/sqrt_probe.py also shows raw v_sqrt_f32 alone is off by an ulp on 15 % of them).  The
// packed field's d2 (< 65536) and the rect records' d2 (< 2^21 on maps up to 1024 cells) are such integers.
// _nz: x = 0 gives NaN (0 * inf); callers that may see 0 and use the value take sqrt_small_int.
__device__ __forceinline__ float sqrt_small_int_nz(float x) {
#ifdef NAVSIM_DIAG_RAW_SQRT          // diagnostic build only: how much of the step is the sqrt correction?
    return __builtin_amdgcn_sqrtf(x);
#endif
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y;
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r * 0.5f, y, s);
}
__device__ __forceinline__ float sqrt_small_int(float x) {
    const float s = sqrt_small_int_nz(x);
    return x == 0.0f ? 0.0f : s;
}

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t hash4(uint64_t seed, uint64_t a, uint64_t b, uint64_t c) {
    uint64_t h = mix64(seed ^ 0x6E6176676D796DULL);
    h = mix64(h ^ a);
    h = mix64(h ^ b);
    h = mix64(h ^ c);
    return h;
}

}  // namespace nv
