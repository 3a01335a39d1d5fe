/*
 * pt_detmath.h — the numerics contract of the render path.
 *
 * The reference calls CUDA's sinf/cosf/acosf/atan2f/expf/logf/powf/log2f
 * (pathtracer.cu:436,439,562-579; envmap_cuda.cu:13-14,70-73; disney.cuh:51-52,
 * 81-82,129,139-142,215; phase_function.cu:6,19; tonemapping.cuh:36,55).  Those
 * are neither glibc's nor ROCm ocml's, and the three libraries disagree by ulps,
 * which forks paths at every discrete decision (lobe choice, roulette, alpha).
 *
 * This header pins each transcendental to one algorithm made only of IEEE-754
 * double add/sub/mul/div, float sqrt, floor and integer bit operations, with
 * contraction disabled.  Any conforming compiler, on host or on gfx950,
 * produces the same bits.  Accuracy is better than 1 ulp of the float result
 * on the ranges the renderer uses (checked against libm in the test-suite).
 *
 * Usable from C++ host code and from HIP device code.  C linkage is not needed:
 * everything is static inline.
 */
#ifndef PT_DETMATH_H
#define PT_DETMATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define PT_HD __host__ __device__ static inline
#else
#define PT_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#define PT_DM_PI      3.14159265358979323846
#define PT_DM_PI_2    1.57079632679489661923
#define PT_DM_PI_4    0.78539816339744830962
#define PT_DM_LN2     0.69314718055994530942
#define PT_DM_INV_LN2 1.44269504088896340736

PT_HD uint64_t pt_dm_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
PT_HD double pt_dm_from_bits(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
PT_HD uint32_t pt_dm_fbits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
PT_HD float pt_dm_ffrom_bits(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
PT_HD int pt_dm_isnan(double x) { return x != x; }
PT_HD double pt_dm_nan(void) { return pt_dm_from_bits(0x7ff8000000000000ull); }
PT_HD double pt_dm_inf(void) { return pt_dm_from_bits(0x7ff0000000000000ull); }

/* Double sqrt built from the float sqrt (IEEE, correctly rounded on host and
 * device) and Newton steps in double, so it does not depend on how a backend
 * expands a native f64 sqrt. Deterministic; within 1 ulp. Input finite. */
PT_HD double pt_dm_sqrt(double v)
{
    if (!(v > 0.0)) return v == 0.0 ? 0.0 : pt_dm_nan();
    double post = 1.0;
    if (v < 1.0e-290) { v = v * 1.3407807929942597e154; post = 8.636168555094445e-78; } /* 2^512, 2^-256 */
    uint64_t b = pt_dm_bits(v);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    int h = e / 2;                      /* truncation toward zero */
    double m = pt_dm_from_bits(b - ((uint64_t)(int64_t)(2 * h) << 52)); /* v * 2^(-2h), in [0.5,4) */
    double s = (double)__builtin_sqrtf((float)m);
    s = 0.5 * (s + m / s);
    s = 0.5 * (s + m / s);
    double r = m - s * s;
    s = s + r / (2.0 * s);
    return pt_dm_from_bits(pt_dm_bits(s) + ((uint64_t)(int64_t)h << 52)) * post;
}

/* sin and cos of a double argument, |x| <= 1e9. */
PT_HD void pt_dm_sincos(double x, double* sn, double* cs)
{
    if (!(x >= -1.0e9 && x <= 1.0e9)) {
        if (pt_dm_isnan(x) || x > 1.0e300 || x < -1.0e300) { *sn = pt_dm_nan(); *cs = pt_dm_nan(); }
        else { *sn = 0.0; *cs = 1.0; }
        return;
    }
    double kd = __builtin_floor(x * 0.63661977236758134308 + 0.5);
    long long k = (long long)kd;
    double r = (x - kd * 1.5707963267948966) - kd * 6.123233995736766e-17;
    double r2 = r * r;
    double sp = r + r * r2 * (-1.0 / 6.0 + r2 * (1.0 / 120.0 + r2 * (-1.0 / 5040.0 + r2 * (1.0 / 362880.0 +
                r2 * (-1.0 / 39916800.0 + r2 * (1.0 / 6227020800.0))))));
    double cp = 1.0 + r2 * (-0.5 + r2 * (1.0 / 24.0 + r2 * (-1.0 / 720.0 + r2 * (1.0 / 40320.0 +
                r2 * (-1.0 / 3628800.0 + r2 * (1.0 / 479001600.0 + r2 * (-1.0 / 87178291200.0)))))));
    switch ((int)(k & 3)) {
    case 0: *sn = sp;  *cs = cp;  break;
    case 1: *sn = cp;  *cs = -sp; break;
    case 2: *sn = -sp; *cs = -cp; break;
    default: *sn = -cp; *cs = sp; break;
    }
}

/* exp of a double, argument clamped to [-700,700]; NaN propagates. */
PT_HD double pt_dm_exp(double x)
{
    if (pt_dm_isnan(x)) return x;
    if (x > 700.0) x = 700.0;
    if (x < -700.0) x = -700.0;
    double kd = __builtin_floor(x * PT_DM_INV_LN2 + 0.5);
    double r = (x - kd * 0.6931471803691238) - kd * 1.9082149292705877e-10;
    double p = 1.0 + r * (1.0 + r * (0.5 + r * (1.0 / 6.0 + r * (1.0 / 24.0 + r * (1.0 / 120.0 + r * (1.0 / 720.0 +
               r * (1.0 / 5040.0 + r * (1.0 / 40320.0 + r * (1.0 / 362880.0 + r * (1.0 / 3628800.0 +
               r * (1.0 / 39916800.0 + r * (1.0 / 479001600.0))))))))))));
    long long k = (long long)kd;                 /* |k| <= 1011 */
    /* split the scale so neither factor leaves the normal range */
    long long k1 = k / 2, k2 = k - k1;
    double s1 = pt_dm_from_bits((uint64_t)(k1 + 1023) << 52);
    double s2 = pt_dm_from_bits((uint64_t)(k2 + 1023) << 52);
    return (p * s1) * s2;
}

/* natural log split as e*ln2 + m, for finite x > 0 (normal or subnormal double). */
PT_HD void pt_dm_log_parts(double x, double* e_out, double* m_out)
{
    uint64_t b = pt_dm_bits(x);
    int e = (int)((b >> 52) & 0x7ff);
    if (e == 0) {                       /* subnormal double: renormalise */
        x = x * 18014398509481984.0;    /* 2^54 */
        b = pt_dm_bits(x);
        e = (int)((b >> 52) & 0x7ff) - 54;
    }
    e -= 1023;
    double m = pt_dm_from_bits((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = 2.0 * s * (1.0 + s2 * (1.0 / 3.0 + s2 * (1.0 / 5.0 + s2 * (1.0 / 7.0 + s2 * (1.0 / 9.0 +
               s2 * (1.0 / 11.0 + s2 * (1.0 / 13.0 + s2 * (1.0 / 15.0 + s2 * (1.0 / 17.0)))))))));
    *e_out = (double)e;
    *m_out = p;
}

PT_HD double pt_dm_log(double x)
{
    if (pt_dm_isnan(x)) return x;
    if (x < 0.0) return pt_dm_nan();
    if (x == 0.0) return -pt_dm_inf();
    if (x > 1.0e308) return x > 1.7976931348623157e308 ? pt_dm_inf() : 709.782712893384;
    double e, m;
    pt_dm_log_parts(x, &e, &m);
    return e * PT_DM_LN2 + m;
}

/* pow for the cases the renderer meets: any finite/inf x, finite y.
 * Negative x yields NaN (the reference only raises to non-integer powers). */
PT_HD double pt_dm_pow(double x, double y)
{
    if (y == 0.0) return 1.0;
    if (x == 1.0) return 1.0;
    if (pt_dm_isnan(x) || pt_dm_isnan(y)) return pt_dm_nan();
    if (x < 0.0) return pt_dm_nan();
    if (x == 0.0) return y > 0.0 ? 0.0 : pt_dm_inf();
    if (x > 1.7976931348623157e308) return y > 0.0 ? pt_dm_inf() : 0.0;
    return pt_dm_exp(y * pt_dm_log(x));
}

/* atan of z >= 0 (inf allowed). */
PT_HD double pt_dm_atan_pos(double z)
{
    double base = 0.0, sign = 1.0;
    if (z > 1.0) { z = 1.0 / z; base = PT_DM_PI_2; sign = -1.0; }   /* atan z = pi/2 - atan(1/z) */
    double add = 0.0;
    if (z > 0.41421356237309503) { add = PT_DM_PI_4; z = (z - 1.0) / (z + 1.0); }
    double z2 = z * z;
    double p = z * (1.0 + z2 * (-1.0 / 3.0 + z2 * (1.0 / 5.0 + z2 * (-1.0 / 7.0 + z2 * (1.0 / 9.0 + z2 * (-1.0 / 11.0 +
               z2 * (1.0 / 13.0 + z2 * (-1.0 / 15.0 + z2 * (1.0 / 17.0 + z2 * (-1.0 / 19.0 + z2 * (1.0 / 21.0 +
               z2 * (-1.0 / 23.0 + z2 * (1.0 / 25.0 + z2 * (-1.0 / 27.0 + z2 * (1.0 / 29.0)))))))))))))));
    double a = add + p;
    return base + sign * a;
}

/* ---- float entry points (the ones the render path calls) ---- */

/* fminf / fmaxf with the NaN rule of IEEE minNum/maxNum (the non-NaN operand wins) and a fixed
 * choice where IEEE leaves one open (equal operands, signed zeros: the second operand). */
PT_HD float pt_fminf(float a, float b) { return (a < b || b != b) ? a : b; }
PT_HD float pt_fmaxf(float a, float b) { return (a > b || b != b) ? a : b; }

PT_HD float pt_sinf(float x) { double s, c; pt_dm_sincos((double)x, &s, &c); return (float)s; }
PT_HD float pt_cosf(float x) { double s, c; pt_dm_sincos((double)x, &s, &c); return (float)c; }
PT_HD float pt_expf(float x) { double xd = (double)x; if (xd > 150.0) xd = 150.0; if (xd < -150.0) xd = -150.0; return (float)pt_dm_exp(xd); }
PT_HD float pt_logf(float x) { return (float)pt_dm_log((double)x); }

PT_HD float pt_log2f(float x)
{
    double xd = (double)x;
    if (pt_dm_isnan(xd)) return x;
    if (xd < 0.0) return (float)pt_dm_nan();
    if (xd == 0.0) return (float)(-pt_dm_inf());
    if (xd > 3.5e38) return (float)pt_dm_inf();
    double e, m;
    pt_dm_log_parts(xd, &e, &m);
    return (float)(e + m * PT_DM_INV_LN2);
}

PT_HD float pt_powf(float x, float y) { return (float)pt_dm_pow((double)x, (double)y); }

PT_HD float pt_acosf(float x)
{
    double xd = (double)x;
    if (!(xd >= -1.0 && xd <= 1.0)) return (float)pt_dm_nan();
    if (xd == -1.0) return (float)PT_DM_PI;
    return (float)(2.0 * pt_dm_atan_pos(pt_dm_sqrt((1.0 - xd) / (1.0 + xd))));
}

PT_HD float pt_atan2f(float y, float x)
{
    double yd = (double)y, xd = (double)x;
    if (pt_dm_isnan(yd) || pt_dm_isnan(xd)) return (float)pt_dm_nan();
    int yneg = (int)(pt_dm_fbits(y) >> 31), xneg = (int)(pt_dm_fbits(x) >> 31);
    double ay = yneg ? -yd : yd, ax = xneg ? -xd : xd;
    double a;
    if (ay == 0.0) a = xneg ? PT_DM_PI : 0.0;
    else if (ax > 1.0e300 && ay > 1.0e300) a = xneg ? 3.0 * PT_DM_PI_4 : PT_DM_PI_4;
    else {
        a = pt_dm_atan_pos(ay / ax);    /* ax == 0 gives +inf -> pi/2 */
        if (xneg && ax != 0.0) a = PT_DM_PI - a;
    }
    return (float)(yneg ? -a : a);
}

#endif /* PT_DETMATH_H */
