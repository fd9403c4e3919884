/* navmath_ref.h -- TEST INFRASTRUCTURE (CPU oracle).  Not part of the product.
 *
 * Deterministic elementary functions used by the oracle.  The reference calls libm
 * (cosf/sinf inside range_libc, np.cos/np.sin/np.arctan2 in keti_robot.py:64-93, human.py:32-41,
 * utils.py:5-9); libm results differ in the last bit between platforms, and one ulp in a beam
 * direction can move a sphere-traced hit by a whole cell (SURVEY.md section 7 "hard parts").
 * DESIGN.md section 4 therefore SPECIFIES these functions operation by operation (Cody-Waite
 * reduction + the classic fdlibm minimax polynomials, IEEE double ops only, no FMA contraction),
 * so that the HIP kernels, which implement the same specification independently in
 * nav-gym_amd/csrc/navmath.hpp, produce bit-identical values.  tests/test_navmath.py checks this
 * file against numpy to <= 2 ulp and (on the GPU) against the device implementation bit for bit.
 *
 * Compile with -ffp-contract=off.
 */
#ifndef NAVMATH_REF_H
#define NAVMATH_REF_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#define NVR_PI      3.14159265358979311600e+00
#define NVR_TWO_PI  6.28318530717958623200e+00   /* == 2*np.pi in float64 */

static inline double nvr_from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t nvr_to_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

/* argument reduction: x = n*(pi/2) + (y0 + y1), |y0| <= ~pi/4; valid for |x| < 1e5 */
static inline int nvr_rem_pio2(double x, double* y0, double* y1) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1  = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double pio2_2  = 6.07710050630396597660e-11;  /* next 33 bits */
    const double pio2_2t = 2.02226624879595063154e-21;  /* pi/2 - pio2_1 - pio2_2 */
    double fn = rint(x * invpio2);
    double r  = x - fn * pio2_1;      /* exact */
    double r2 = r - fn * pio2_2;
    double w  = fn * pio2_2t;
    double a  = r2 - w;
    *y0 = a;
    *y1 = (r2 - a) - w;
    return (int)((long long)fn & 3);
}

static inline double nvr_ksin(double x, double y) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

static inline double nvr_kcos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double hz = 0.5 * z;
    double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

static inline void nvr_sincos(double x, double* s, double* c) {
    double y0, y1;
    int n = nvr_rem_pio2(x, &y0, &y1);
    double sn = nvr_ksin(y0, y1);
    double cs = nvr_kcos(y0, y1);
    switch (n) {
        case 0:  *s = sn;  *c = cs;  break;
        case 1:  *s = cs;  *c = -sn; break;
        case 2:  *s = -sn; *c = -cs; break;
        default: *s = -cs; *c = sn;  break;
    }
}
static inline double nvr_sin(double x) { double s, c; nvr_sincos(x, &s, &c); return s; }
static inline double nvr_cos(double x) { double s, c; nvr_sincos(x, &s, &c); return c; }

static inline double nvr_atan(double x) {
    static const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01,
                                     9.82793723247329054082e-01, 1.57079632679489655800e+00};
    static const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17,
                                     1.39033110312309984516e-17, 6.12323399573676603587e-17};
    static const double aT[11] = {
        3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01,
        -1.11111104054623557880e-01, 9.09088713343650656196e-02, -7.69187620504482999495e-02,
        6.66107313738753120669e-02, -5.83357013379057348645e-02, 4.97687799461593236017e-02,
        -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    int neg = x < 0.0;
    double ax = fabs(x);
    int id;
    if (ax >= 7.3786976294838206464e19) {       /* 2^66 */
        double z = atanhi[3] + atanlo[3];
        return neg ? -z : z;
    }
    if (ax < 0.4375) {
        if (ax < 1.862645149230957e-09) return x; /* 2^-29 */
        id = -1;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) { id = 0; ax = (2.0 * ax - 1.0) / (2.0 + ax); }
        else             { id = 1; ax = (ax - 1.0) / (ax + 1.0); }
    } else {
        if (ax < 2.4375) { id = 2; ax = (ax - 1.5) / (1.0 + 1.5 * ax); }
        else             { id = 3; ax = -1.0 / ax; }
    }
    double z = ax * ax;
    double w = z * z;
    double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) {
        double r = ax - ax * (s1 + s2);
        return neg ? -r : r;
    }
    double r = atanhi[id] - ((ax * (s1 + s2) - atanlo[id]) - ax);
    return neg ? -r : r;
}

/* atan2 for finite arguments; (0,0) -> 0 */
static inline double nvr_atan2(double y, double x) {
    const double pi_lo = 1.2246467991473531772e-16;
    const double pi_o_2 = 1.5707963267948965580e+00;
    if (x == 0.0 && y == 0.0) return 0.0;
    if (x == 0.0) return (y < 0.0) ? -pi_o_2 : pi_o_2;
    if (y == 0.0) return (x < 0.0) ? NVR_PI : 0.0;
    double z = nvr_atan(fabs(y / x));
    if (x > 0.0) return (y < 0.0) ? -z : z;
    return (y < 0.0) ? (z - pi_lo) - NVR_PI : NVR_PI - (z - pi_lo);
}

/* exp for x <= 0 (the social-force decays); x < -700 -> 0 */
static inline double nvr_exp_neg(double x) {
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    if (x > 0.0) x = 0.0;
    if (x < -700.0) return 0.0;
    double fk = rint(x * invln2);
    double hi = x - fk * ln2HI;
    double lo = fk * ln2LO;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    int k = (int)fk;                       /* -1010 <= k <= 0 */
    return y * nvr_from_bits((uint64_t)(k + 1023) << 52);
}

/* Python float % (2*pi): fmod then sign fix-up (CPython float_rem) */
static inline double nvr_mod_2pi(double x) {
    double m = fmod(x, NVR_TWO_PI);
    if (m != 0.0 && m < 0.0) m += NVR_TWO_PI;
    return m;
}

/* utils.py:5-9 angle_correction */
static inline double nvr_wrap_pi(double a) {
    double s, c;
    nvr_sincos(a, &s, &c);
    return nvr_atan2(s, c);
}

static inline uint64_t nvr_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t nvr_hash4(uint64_t seed, uint64_t a, uint64_t b, uint64_t c) {
    uint64_t h = nvr_mix64(seed ^ 0x6E6176676D796DULL);
    h = nvr_mix64(h ^ a);
    h = nvr_mix64(h ^ b);
    h = nvr_mix64(h ^ c);
    return h;
}

#endif /* NAVMATH_REF_H */
