// jk_math2.h -- the extra arithmetic of the PacBio path (reference: src/hts_pacbio.h, src/hts_pacbio.cpp).
//
// The reference's per-read set-up calls libstdc++ (lognormal_distribution, chi_squared_distribution),
// glibc (exp, pow, log, log10, sqrt), R's nmath (qchisq, pnorm5, qnorm5; NOT in /root/reference: R itself,
// version unpinned by DESCRIPTION:26) and x87 long double (runif_ab, src/pcg.h:103-105).  For the GPU to
// reproduce the oracle's bytes every one of them is restated here with a fixed operation order:
//   * jk_exp / jk_pow : glibc 2.35 exp/pow, FMA ifunc variants, transcribed from libm's __exp_fma /
//     __pow_fma (tables lifted by tools/extract_glibc_log_table.py); jk_pow covers the main path only
//     (x a positive normal, 2^-65 <= |y| < 2^63, result far from overflow/underflow) and reports
//     anything else through *ok.
//   * jk_log10        : glibc's e_log10.c on top of jk_log.
//   * jk_qnorm / jk_pnorm : Wichura's AS 241 (PPND16) and Cody's 1969 rational approximations, the
//     published algorithms R's qnorm5 / pnorm5 implement.  Parity with R itself is UNPINNED (R is not
//     available here); parity between the CPU oracle and the GPU is by construction.
//   * jk_x87_*        : the 64-bit-significand arithmetic of `a + runif_01 * (b - a)`.
#pragma once
#include "jk_math.h"

static constexpr double JK_EXP_C[8] = JK_EXP_CONSTS;
static constexpr uint64_t JK_EXP_T[256] = JK_EXP_TAB;
static constexpr double JK_PLOG_A[7] = JK_POWLOG_POLY;
static constexpr double JK_PLOG_T[384] = JK_POWLOG_TAB;     // {invc, logc, logctail} x 128

// glibc's specialcase() of exp / pow (sysdeps/ieee754/dbl-64/e_exp.c, e_pow.c): 512 <= |x| < 1024, where the scale
// 2^k alone would over- or underflow.  libm's FMA build fuses the sum of the k > 0 branch and leaves those of the k < 0
// branch unfused (found and checked bit for bit against this image's libm over the whole range,
// tests/test_host_primitives.py).
JK_HD double jk_exp_specialcase(double tmp, uint64_t sbits, uint64_t ki) {
    if ((ki & 0x80000000ULL) == 0) {
        sbits -= 1009ULL << 52;                      // k > 0: the exponent of scale might have overflowed by <= 460
        const double scale = jk_u2d(sbits);
        return 0x1p1009 * __builtin_fma(scale, tmp, scale);
    }
    sbits += 1022ULL << 52;                          // k < 0: special care in the subnormal range
    const double scale = jk_u2d(sbits);
    double y = scale + scale * tmp;
    if (__builtin_fabs(y) < 1.0) {
        double one = 1.0;
        if (y < 0.0) one = -1.0;
        double lo = scale - y + scale * tmp;
        const double hi = one + y;
        lo = one - hi + y + lo;
        y = (hi + lo) - one;
        if (y == 0.0) y = jk_u2d(sbits & 0x8000000000000000ULL);
    }
    return 0x1p-1022 * y;
}

// Core shared by exp and pow: exp(x + xtail).  (Always true now: kept as a status for callers that distinguish.)
JK_HD bool jk_exp_core(double x, double xtail, double* out) {
    uint32_t abstop = (uint32_t)(jk_d2u(x) >> 52) & 0x7ffu;
    bool special = false;
    if (abstop - 0x3c9u >= 0x3fu) {
        if (abstop - 0x3c9u >= 0x80000000u) { *out = 1.0 + x; return true; }   // |x| < 2^-54
        if (abstop >= 0x409u) {                                                // |x| >= 1024 (inf / nan are the callers' business)
            *out = (jk_d2u(x) >> 63) ? 0.0 : __builtin_inf();                  // __math_uflow(0) / __math_oflow(0)
            return true;
        }
        special = true;                                                        // 512 <= |x| < 1024
    }
    const double InvLn2N = JK_EXP_C[0], Shift = JK_EXP_C[1], NegLn2hiN = JK_EXP_C[2], NegLn2loN = JK_EXP_C[3];
    const double C2 = JK_EXP_C[4], C3 = JK_EXP_C[5], C4 = JK_EXP_C[6], C5 = JK_EXP_C[7];
    const double z = __builtin_fma(x, InvLn2N, Shift);
    const uint64_t ki = jk_d2u(z);
    const double kd = z - Shift;
    double r = __builtin_fma(kd, NegLn2hiN, x);
    r = __builtin_fma(kd, NegLn2loN, r);
    r = xtail + r;
    const uint32_t idx = 2u * (uint32_t)(ki & 127u);
    const uint64_t top = ki << 45;
    const double tail = jk_u2d(JK_EXP_T[idx]);
    const uint64_t sbits = JK_EXP_T[idx + 1] + top;
    const double p = __builtin_fma(r, C3, C2);
    const double t = r + tail;
    const double r2 = r * r;
    const double q = __builtin_fma(r, C5, C4);
    const double pt = __builtin_fma(p, r2, t);
    const double r4 = r2 * r2;
    const double tmp = __builtin_fma(q, r4, pt);
    if (special) { *out = jk_exp_specialcase(tmp, sbits, ki); return true; }
    const double scale = jk_u2d(sbits);
    *out = __builtin_fma(tmp, scale, scale);
    return true;
}

// glibc exp(double).  In exp itself xtail does not exist (r is not touched): jk_exp_core(x, 0) adds +0.0,
// which changes nothing except -0.0 -> +0.0 of r, and r = -0.0 cannot occur (x = 0 takes the tiny branch).
JK_HD bool jk_exp(double x, double* out) { return jk_exp_core(x, 0.0, out); }

// glibc pow(x, y), main path.  *ok = false when the arguments need glibc's special handling.
JK_HD double jk_pow(double x, double y, bool* ok) {
    const uint64_t ix = jk_d2u(x), iy = jk_d2u(y);
    const uint32_t topx = (uint32_t)(ix >> 52), topy = (uint32_t)(iy >> 52);
    // pow(+0, y) = +0 for finite y > 0 (glibc's zero-base branch returns x * x): an event probability of exactly 0
    if (ix == 0 && !(iy >> 63) && (iy << 1) != 0 && (topy & 0x7ffu) != 0x7ffu) return 0.0;
    if (topx - 1u > 0x7fdu || (topy & 0x7ffu) - 0x3beu > 0x7fu) { *ok = false; return 0.0; }
    // log_inline
    const uint64_t tmp = ix - 0x3fe6955500000000ULL;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ULL);
    const double z = jk_u2d(iz);
    const double kd = (double)k;
    const double invc = JK_PLOG_T[3 * i], logc = JK_PLOG_T[3 * i + 1], logctail = JK_PLOG_T[3 * i + 2];
    const double* A = JK_PLOG_A;
    const double r = __builtin_fma(z, invc, -1.0);
    const double t1 = __builtin_fma(kd, JK_LOG_LN2HI, logc);
    const double ar = A[0] * r;
    const double lo1 = __builtin_fma(kd, JK_LOG_LN2LO, logctail);
    const double p12 = __builtin_fma(r, A[2], A[1]);
    const double p34 = __builtin_fma(r, A[4], A[3]);
    const double t2 = t1 + r;
    const double ar2 = r * ar;
    const double d12 = t1 - t2;
    const double ar3 = r * ar2;
    const double lo3 = __builtin_fma(ar, r, -ar2);
    const double lo2 = d12 + r;
    const double p56 = __builtin_fma(r, A[6], A[5]);
    const double hi = t2 + ar2;
    const double d2h = t2 - hi;
    const double p36 = __builtin_fma(p56, ar2, p34);
    const double lo4 = d2h + ar2;
    const double p = __builtin_fma(ar2, p36, p12);
    double lo = lo1 + lo2;
    lo = lo + lo3;
    lo = lo + lo4;
    lo = __builtin_fma(ar3, p, lo);
    const double yhi = hi + lo;
    double ylo = hi - yhi;
    ylo = ylo + lo;
    // exp_inline(y * log(x))
    const double ehi = y * yhi;
    double elo = __builtin_fma(yhi, y, -ehi);
    elo = __builtin_fma(y, ylo, elo);
    double res;
    *ok = jk_exp_core(ehi, elo, &res);
    // the tiny branch of exp_inline returns 1.0 + x like exp's; with xtail it is the same value
    return res;
}

// glibc __ieee754_log10 (sysdeps/ieee754/dbl-64/e_log10.c), x > 0 finite
JK_HD double jk_log10(double x) {
    uint64_t ix = jk_d2u(x);
    int32_t hx = (int32_t)(ix >> 32);
    int32_t k = 0;
    if (hx < 0x00100000) {                       // subnormal (zero / negative are not used by the path)
        if (((ix << 1) == 0)) return -__builtin_inf();
        if (hx < 0) return __builtin_nan("");
        k -= 54; x *= 0x1p54; ix = jk_d2u(x); hx = (int32_t)(ix >> 32);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    const int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
    hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
    const double y = (double)(k + i);
    x = jk_u2d(((uint64_t)(uint32_t)hx << 32) | (ix & 0xffffffffULL));
    const double ivln10 = 4.34294481903251816668e-01, log10_2hi = 3.01029995663611771306e-01,
                 log10_2lo = 3.69423907715893078616e-13;
    const double z = y * log10_2lo + ivln10 * jk_log(x);
    return z + y * log10_2hi;
}

// ---- normal quantile: AS 241 (Wichura 1988), PPND16, as used by R's qnorm5(p, 0, 1, lower, !log) ----
JK_HD double jk_qnorm(double p) {
    if (!(p > 0.0)) return -__builtin_inf();
    if (!(p < 1.0)) return __builtin_inf();
    const double q = p - 0.5;
    double r, val;
    if (__builtin_fabs(q) <= 0.425) {
        r = .180625 - q * q;
        val = q * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r
                      + 45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r
                    + 133.14166789178437745) * r + 3.387132872796366608)
              / (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r
                     + 21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r
                   + 42.313330701600911252) * r + 1.);
        return val;
    }
    r = (q < 0) ? p : (0.5 - p + 0.5);
    r = jk_sqrt(-jk_log(r));
    if (r <= 5.) {
        r += -1.6;
        val = (((((((r * 7.7454501427834140764e-4 + .0227238449892691845833) * r + .24178072517745061177) * r
                   + 1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r
                + 4.6303378461565452959) * r + 1.42343711074968357734)
              / (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + .0151986665636164571966) * r
                     + .14810397642748007459) * r + .68976733498510000455) * r + 1.6763848301838038494) * r
                  + 2.05319162663775882187) * r + 1.);
    } else {
        r += -5.;
        val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r + .0012426609473880784386) * r
                   + .026532189526576123093) * r + .29656057182850489123) * r + 1.7848265399172913358) * r
                + 5.4637849111641143699) * r + 6.6579046435011037772)
              / (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r
                     + 7.868691311456132591e-4) * r + .0148753612908506148525) * r + .13692988092273580531) * r
                  + .59983220655588793769) * r + 1.);
    }
    return (q < 0.0) ? -val : val;
}

// ---- x87 64-bit-significand helpers --------------------------------------------------------------
// A positive extended value is m * 2^e with m in [2^63, 2^64) (or m = 0).
struct jk_x87 { uint64_t m; int32_t e; };

JK_HD jk_x87 jk_x87_norm128(jk_u128 v, int32_t e) {      // round-to-nearest-even of v * 2^e to 64 bits
    jk_x87 r; r.m = 0; r.e = 0;
    if (v == 0) return r;
    const uint64_t hi = (uint64_t)(v >> 64);
    const int lz = hi ? jk_clz64(hi) : 64 + jk_clz64((uint64_t)v);
    const int sh = 64 - lz;                               // bits to drop (may be <= 0)
    if (sh <= 0) { r.m = (uint64_t)(v << (-sh)); r.e = e + sh; return r; }
    const jk_u128 half = (jk_u128)1 << (sh - 1), rem = v & (((jk_u128)1 << sh) - 1);
    jk_u128 kept = v >> sh;
    if (rem > half || (rem == half && (kept & 1))) kept++;
    if (kept >> 64) { kept >>= 1; e++; }                  // carried out to 2^64
    r.m = (uint64_t)kept; r.e = e + sh;
    return r;
}
JK_HD jk_x87 jk_x87_from_double(double d) {                // exact (d > 0, normal)
    const uint64_t b = jk_d2u(d);
    jk_x87 r;
    r.m = ((b & 0xfffffffffffffULL) | 0x10000000000000ULL) << 11;
    r.e = (int32_t)((b >> 52) & 0x7ff) - 1075 - 11;
    return r;
}
JK_HD jk_x87 jk_x87_mul(jk_x87 a, jk_x87 b) { return jk_x87_norm128((jk_u128)a.m * b.m, a.e + b.e); }
JK_HD jk_x87 jk_x87_add(jk_x87 a, jk_x87 b) {            // positive operands
    if (a.m == 0) return b;
    if (b.m == 0) return a;
    if (a.e < b.e) { jk_x87 t = a; a = b; b = t; }
    const int d = a.e - b.e;                               // >= 0
    // exact sum in 128+ bits is only needed while b can affect rounding: beyond 66 bits it is a sticky bit
    jk_u128 A = (jk_u128)a.m << 63;                        // a.m * 2^63, scale 2^(a.e - 63)
    jk_u128 B;
    if (d <= 63) B = (jk_u128)b.m << (63 - d);
    else {                                                 // b only matters as a sticky contribution
        const int sft = d - 63;
        if (sft >= 64) B = 1;
        else { B = b.m >> sft; if (b.m & ((1ULL << sft) - 1)) B |= 1; }
    }
    return jk_x87_norm128(A + B, a.e - 63);
}
JK_HD double jk_x87_to_double(jk_x87 a) {                  // RNE to binary64 (normal range)
    if (a.m == 0) return 0.0;
    uint64_t kept = a.m >> 11;
    const uint64_t rem = a.m & 0x7ff;
    int32_t e = a.e + 11;
    if (rem > 0x400 || (rem == 0x400 && (kept & 1))) { kept++; if (kept >> 53) { kept >>= 1; e++; } }
    return jk_u2d(((uint64_t)(e + 1075) << 52) | (kept & 0xfffffffffffffULL));
}
// (double)(a + runif_01 * c) for raw draw x, with a and c given as x87 values (c = b - a computed by the
// host in real long double): runif_ab, src/pcg.h:103-105, then the conversion at the call site.
JK_HD double jk_runif_ab(uint64_t x, jk_x87 a, jk_x87 c) {
    jk_x87 r;                                              // runif_01 = (x + 1) * 2^-64 exactly
    if (x == ~0ULL) { r.m = 1ULL << 63; r.e = -63; }
    else { const uint64_t v = x + 1; const int lz = jk_clz64(v); r.m = v << lz; r.e = -64 - lz; }
    return jk_x87_to_double(jk_x87_add(a, jk_x87_mul(r, c)));
}
