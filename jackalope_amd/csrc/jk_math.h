// jk_math.h -- the arithmetic the read generators share between the HIP kernels and the host
// set-up code (every function is __host__ __device__; the library is built with hipcc only).
//
// Everything here is integer / IEEE-754 binary64 arithmetic chosen to reproduce, bit for bit, what
// the reference computes on x86-64 with x87 `long double`, libstdc++ 11 and glibc 2.35:
//   * pcg64                       /root/reference/inst/include/pcg/pcg_random.hpp:352-477,971-998,1675
//   * runif_01 and its consumers  /root/reference/src/pcg.h:99-105 and call sites cited below
//   * std::gamma_distribution     /usr/include/c++/11/bits/random.tcc:2330-2392 (via hts_illumina.h:301)
//   * glibc log(double)           sysdeps/ieee754/dbl-64/e_log.c, FMA ifunc variant
// Compile with -ffp-contract=off: every fma below is explicit and every other operation must round
// on its own, as in the reference's -O2 x86-64 build.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "jk_log_data.h"

#define JK_HD __host__ __device__ __forceinline__

typedef unsigned __int128 jk_u128;

// ---------------------------------------------------------------------------------------------
// pcg64 (setseq_xsl_rr_128_64).  State kept as two 64-bit halves so it lives in 4 + 4 VGPRs.
// ---------------------------------------------------------------------------------------------
#define JK_PCG_MULT_HI 2549297995355413924ULL
#define JK_PCG_MULT_LO 4865540595714422341ULL

struct jk_pcg64 {
    uint64_t s_hi, s_lo, inc_hi, inc_lo;
};

JK_HD jk_u128 jk_mk128(uint64_t hi, uint64_t lo) { return ((jk_u128)hi << 64) | lo; }

// engine(state, stream) ctor: inc = (stream << 1) | 1; state = (seed + inc) * MULT + inc
// (pcg_random.hpp:471-477 with bump() :387-390).  `w` = the 8 32-bit sub-seeds of
// src/pcg.h:48-61 (fill_seeds): seed1 = w0:w1:w2:w3, seed2 = w4:w5:w6:w7.
JK_HD jk_pcg64 jk_pcg_seed(const uint32_t* w) {
    jk_u128 seed1 = jk_mk128(((uint64_t)w[0] << 32) + w[1], ((uint64_t)w[2] << 32) + w[3]);
    jk_u128 seed2 = jk_mk128(((uint64_t)w[4] << 32) + w[5], ((uint64_t)w[6] << 32) + w[7]);
    jk_u128 mult = jk_mk128(JK_PCG_MULT_HI, JK_PCG_MULT_LO);
    jk_u128 inc = (seed2 << 1) | 1;
    jk_u128 st = (seed1 + inc) * mult + inc;
    jk_pcg64 e;
    e.s_hi = (uint64_t)(st >> 64); e.s_lo = (uint64_t)st;
    e.inc_hi = (uint64_t)(inc >> 64); e.inc_lo = (uint64_t)inc;
    return e;
}

// operator(): advance, then XSL-RR of the NEW state (128-bit engines have output_previous = false).
JK_HD uint64_t jk_pcg_next_ref(jk_pcg64& e) {
    jk_u128 st = jk_mk128(e.s_hi, e.s_lo) * jk_mk128(JK_PCG_MULT_HI, JK_PCG_MULT_LO) + jk_mk128(e.inc_hi, e.inc_lo);
    uint64_t hi = (uint64_t)(st >> 64), lo = (uint64_t)st;
    e.s_hi = hi; e.s_lo = lo;
    unsigned rot = (unsigned)(hi >> 58);
    uint64_t x = hi ^ lo;
    return (x >> rot) | (x << ((64u - rot) & 63u));
}

// The state as four 32-bit limbs: a loop-carried 64-bit value has to live in an aligned register pair, and the new
// limbs come out of different instructions, so 64-bit state costs one or two v_mov per step just to re-pair them.
struct jk_pcg64d {
    uint32_t s0, s1, s2, s3;
    uint64_t inc_lo, inc_hi;
};
JK_HD jk_pcg64d jk_pcg_limbs(const jk_pcg64& e) {
    jk_pcg64d d;
    d.s0 = (uint32_t)e.s_lo; d.s1 = (uint32_t)(e.s_lo >> 32); d.s2 = (uint32_t)e.s_hi; d.s3 = (uint32_t)(e.s_hi >> 32);
    d.inc_lo = e.inc_lo; d.inc_hi = e.inc_hi;
    return d;
}
JK_HD uint64_t jk_pcg_next_ref(jk_pcg64d& d) {
    jk_pcg64 e;
    e.s_lo = ((uint64_t)d.s1 << 32) | d.s0; e.s_hi = ((uint64_t)d.s3 << 32) | d.s2; e.inc_lo = d.inc_lo; e.inc_hi = d.inc_hi;
    const uint64_t r = jk_pcg_next_ref(e);
    d.s0 = (uint32_t)e.s_lo; d.s1 = (uint32_t)(e.s_lo >> 32); d.s2 = (uint32_t)e.s_hi; d.s3 = (uint32_t)(e.s_hi >> 32);
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(JK_PCG_PLAIN)
// The same step written for the gfx950 VALU.  The 128x128->128 multiply-add is the hot spot of every
// generator kernel (~1 206 steps per read pair); the compiler's expansion of the __int128 expression
// spends a third of its ~35 instructions on register moves and on a separate 4-instruction carry chain for
// "+ increment".  Here the increment rides in the 64-bit addend of v_mad_u64_u32 and the two carries that
// can leave bit 63 are taken from the instruction's carry-out (which plain C cannot name):
//     t0 = s0*m0 + (c1:c0)           -> limb 0, carry cA (worth 2^64)
//     t1 = s1*m0 + hi(t0); t1 += s0*m1 -> limb 1, carry cB (worth 2^96)
//     h  = s2*m0 + (c3:c2) + s1*m1 + s0*m2 + hi(t1) + cA, hi(h) += lo32(s0*m3 + s1*m2 + s2*m1 + s3*m0) + cB
// gfx940-family hazard: a VALU result in an SGPR (carry-out / vcc) needs 2 wait states before a VALU reads
// it, and the compiler cannot see inside asm: the block below is ordered so that this always holds.
__device__ __forceinline__ uint64_t jk_pcg_next(jk_pcg64d& e) {
    const uint32_t s0 = e.s0, s1 = e.s1, s2 = e.s2, s3 = e.s3;
    const uint32_t m0 = (uint32_t)JK_PCG_MULT_LO, m1 = (uint32_t)(JK_PCG_MULT_LO >> 32);
    const uint32_t m2 = (uint32_t)JK_PCG_MULT_HI, m3 = (uint32_t)(JK_PCG_MULT_HI >> 32);
    uint64_t t0, t1, h, cA, cB, junk;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t0), "=s"(cA) : "v"(s0), "s"(m0), "v"(e.inc_lo));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(h), "=s"(junk) : "v"(s2), "s"(m0), "v"(e.inc_hi));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t1), "=s"(junk) : "v"(s1), "s"(m0), "v"(t0 >> 32));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(h), "=s"(junk) : "v"(s1), "s"(m1), "v"(h));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t1), "=s"(cB) : "v"(s0), "s"(m1), "v"(t1));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(h), "=s"(junk) : "v"(s0), "s"(m2), "v"(h));
    const uint32_t hu = (uint32_t)(h >> 32) + (s0 * m3 + s1 * m2 + s2 * m1 + s3 * m0);
    const uint32_t n0 = (uint32_t)t0, n1 = (uint32_t)t1;
    uint32_t h_lo, h_hi, x_lo;
    // Wait states by construction: cA is at least two VALU instructions old when the first add reads it (the
    // t1 and h chains depend on t0 and feed this add); cB may have been written by the instruction just
    // before this block, so its add comes third; vcc is read two instructions after it is written.
    asm("v_addc_co_u32_e64 %[hl], vcc, %[h0], %[t1h], %[cA]\n\t"
        "v_xor_b32_e32 %[xl], %[n0], %[hl]\n\t"
        "v_addc_co_u32_e64 %[hh], %[jk], %[hu], 0, %[cB]\n\t"
        "v_addc_co_u32_e32 %[hh], vcc, 0, %[hh], vcc"
        : [hl] "=&v"(h_lo), [hh] "=&v"(h_hi), [xl] "=&v"(x_lo), [jk] "=&s"(junk)
        : [h0] "v"((uint32_t)h), [t1h] "v"((uint32_t)(t1 >> 32)), [cA] "s"(cA), [hu] "v"(hu), [cB] "s"(cB), [n0] "v"(n0)
        : "vcc");
    e.s0 = n0; e.s1 = n1; e.s2 = h_lo; e.s3 = h_hi;
    // XSL-RR: rotate (hi ^ lo) right by the top 6 bits of the state; rot >= 32 swaps the two words
    // (bit-select on the sign of the high word instead of compare + vcc + two selects)
    const uint32_t x_hi = n1 ^ h_hi, rot = h_hi >> 26;
    const uint32_t a = __builtin_amdgcn_alignbit(x_hi, x_lo, rot);      // (x >> (rot & 31)) low word
    const uint32_t b = __builtin_amdgcn_alignbit(x_lo, x_hi, rot);      // ... high word
    uint32_t sw, r_lo, r_hi;
    asm("v_ashrrev_i32_e32 %[sw], 31, %[hh]\n\t"
        "v_bfi_b32 %[rl], %[sw], %[b], %[a]\n\t"
        "v_bfi_b32 %[rh], %[sw], %[a], %[b]"
        : [sw] "=&v"(sw), [rl] "=&v"(r_lo), [rh] "=&v"(r_hi)
        : [hh] "v"(h_hi), [a] "v"(a), [b] "v"(b));
    return ((uint64_t)r_hi << 32) | r_lo;
}
__device__ __forceinline__ uint64_t jk_pcg_next(jk_pcg64& e) {
    jk_pcg64d d = jk_pcg_limbs(e);
    const uint64_t r = jk_pcg_next(d);
    e.s_lo = ((uint64_t)d.s1 << 32) | d.s0; e.s_hi = ((uint64_t)d.s3 << 32) | d.s2;
    return r;
}
__host__ inline uint64_t jk_pcg_next(jk_pcg64& e) { return jk_pcg_next_ref(e); }    // host code seen by the device pass
__host__ inline uint64_t jk_pcg_next(jk_pcg64d& e) { return jk_pcg_next_ref(e); }
#else
JK_HD uint64_t jk_pcg_next(jk_pcg64& e) { return jk_pcg_next_ref(e); }
JK_HD uint64_t jk_pcg_next(jk_pcg64d& e) { return jk_pcg_next_ref(e); }
#endif

// ---------------------------------------------------------------------------------------------
// runif_01(eng) = ((long double)x + 1) / ((long double)(2^64-1) + 2)  (src/pcg.h:99-101).
// On x86-64 the divisor rounds to exactly 2^64 in the 64-bit x87 significand and x + 1 is exact,
// so runif_01 == (x + 1) * 2^-64 exactly.  The consumers below restate each use in integers.
// ---------------------------------------------------------------------------------------------
JK_HD int jk_clz64(uint64_t v) { return __builtin_clzll(v); }

// (uint64)(runif_01 * n): the x87 product (x+1)*n is rounded to nearest-even at 64 significant bits
// and then truncated.  P = (x+1)*n < 2^128; rounding can only change floor(P / 2^64) by carrying
// out of an all-ones (hence odd) kept part, where ties also round up, so adding half of the
// dropped granule and taking the high word is exact.
// Call sites: src/alias_sampler.h:55 (n = table size), src/hts_illumina.h:216 (4.0), :254 (3.0),
// src/hts_pacbio.cpp (4, 3).  n must be < 2^63.
JK_HD uint64_t jk_runif_index(uint64_t x, uint64_t n) {
    jk_u128 P = (jk_u128)x * n + n;
    uint64_t hi = (uint64_t)(P >> 64), lo = (uint64_t)P;
    if (hi == 0) return 0;
    uint64_t half = 1ULL << (63 - jk_clz64(hi));
    return hi + ((lo + half) < lo ? 1ULL : 0ULL);
}

// double u = runif_01: round-to-nearest-even of (x+1) * 2^-64 to binary64
// (src/alias_sampler.h:57, src/hts_illumina.cpp:132,215, src/hts.h:265, src/hts_illumina.h:251).
JK_HD double jk_runif_double(uint64_t x) {
    if (x == ~0ULL) return 1.0;
    return (double)(x + 1) * 0x1p-64;     // u64 -> f64 conversion is RNE; the scaling is exact
}

// libstdc++ generate_canonical<double,53> over a 64-bit URBG: (double)x / 2^64, clamped below 1
// (random.tcc:3348-3380; one engine call because log2(range) = 64 >= 53).
JK_HD double jk_canonical(uint64_t x) {
    double r = (double)x * 0x1p-64;
    return r >= 1.0 ? 0x1.fffffffffffffp-1 : r;
}

// runif_01(eng) < 0.5 compared in long double (src/hts_illumina.cpp:352): (x+1) < 2^63.
JK_HD bool jk_runif_lt_half(uint64_t x) { return x < 0x7fffffffffffffffULL; }

// qint = runif_01 * 10 + '!' truncated to unsigned char (src/hts_illumina.h:238): two x87
// roundings (product, then sum), both to 64 significant bits.
JK_HD uint8_t jk_n_qual(uint64_t x) {
    jk_u128 P = (jk_u128)x * 10 + 10;          // (x+1)*10, scaled by 2^64
    uint64_t hi = (uint64_t)(P >> 64);
    if (hi != 0) {                              // round P to 64 significant bits (nearest even)
        int sh = 64 - jk_clz64(hi);             // bits to drop
        jk_u128 g = (jk_u128)1 << sh, half = g >> 1, rem = P & (g - 1);
        P -= rem;
        if (rem > half || (rem == half && ((P >> sh) & 1))) P += g;
    }
    // sum t + 33 lies in [33, 43] -> ulp 2^-58, i.e. granule 2^6 at this scale; only the integer
    // part survives, and it changes only through a carry out of an all-ones fraction.
    P += (jk_u128)33 << 64;
    P += 32;
    return (uint8_t)(uint64_t)(P >> 64);
}

// frag_start = (uint64)(u * (double)span) with u = (double)runif_01 (src/hts_illumina.cpp:215-216).
JK_HD uint64_t jk_frag_start(uint64_t x, uint64_t span) {
    return (uint64_t)(jk_runif_double(x) * (double)span);
}

// ---------------------------------------------------------------------------------------------
// glibc 2.35 log(double), FMA variant (the one the x86-64 ifunc selects on any FMA+AVX2 CPU).
// Operation order transcribed from libm's `__log_fma`; constants from jk_log_data.h.
// ---------------------------------------------------------------------------------------------
static constexpr double JK_LOG_A[5] = JK_LOG_POLY_A;
static constexpr double JK_LOG_B[11] = JK_LOG_POLY_B;
static constexpr double JK_LOG_T[256] = JK_LOG_TAB;

JK_HD uint64_t jk_d2u(double d) { union { double d; uint64_t u; } c; c.d = d; return c.u; }
JK_HD double jk_u2d(uint64_t u) { union { double d; uint64_t u; } c; c.u = u; return c.d; }

JK_HD double jk_log(double x) {
    uint64_t ix = jk_d2u(x);
    uint32_t top = (uint32_t)(ix >> 48);
    const uint64_t LO = 0x3fee000000000000ULL;   // asuint64(1 - 0x1p-4)
    const uint64_t HI = 0x3ff1090000000000ULL;   // asuint64(1 + 0x1.09p-4)
    if (ix - LO < HI - LO) {
        if (ix == 0x3ff0000000000000ULL) return 0.0;
        const double* B = JK_LOG_B;
        double r = x - 1.0;
        double a = __builtin_fma(r, B[2], B[1]);
        double b = __builtin_fma(r, B[5], B[4]);
        double r2 = r * r;
        double c = __builtin_fma(r, B[8], B[7]);
        a = __builtin_fma(r2, B[3], a);
        b = __builtin_fma(r2, B[6], b);
        double r3 = r * r2;
        c = __builtin_fma(r2, B[9], c);
        c = __builtin_fma(r3, B[10], c);
        c = __builtin_fma(c, r3, b);
        double poly = __builtin_fma(c, r3, a);
        double t = __builtin_fma(r, 0x1p27, r);
        double rhi = __builtin_fma(-0x1p27, r, t);
        double rhi2 = rhi * rhi;
        double rlo = r - rhi;
        double hi = __builtin_fma(rhi2, B[0], r);
        double d = r - hi;
        double rr = r + rhi;
        double lo = __builtin_fma(rhi2, B[0], d);
        double e = B[0] * rlo;
        lo = __builtin_fma(e, rr, lo);
        double y = __builtin_fma(poly, r3, lo);
        return hi + y;
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
        if ((ix << 1) == 0) return -__builtin_inf();                 // log(+-0) = -inf
        if (ix == 0x7ff0000000000000ULL) return x;                   // log(inf) = inf
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return __builtin_nan("");   // x < 0 or NaN
        ix = jk_d2u(x * 0x1p52);                                     // subnormal: normalise
        ix -= 52ULL << 52;
    }
    uint64_t tmp = ix - 0x3fe6000000000000ULL;
    int i = (int)((tmp >> 45) & 127);
    int k = (int)((int64_t)tmp >> 52);
    uint64_t iz = ix - (tmp & 0xfff0000000000000ULL);
    double invc = JK_LOG_T[2 * i], logc = JK_LOG_T[2 * i + 1];
    double z = jk_u2d(iz);
    double kd = (double)k;
    const double* A = JK_LOG_A;
    double r = __builtin_fma(z, invc, -1.0);
    double w = __builtin_fma(kd, JK_LOG_LN2HI, logc);
    double p1 = __builtin_fma(r, A[2], A[1]);
    double hi = r + w;
    double r2 = r * r;
    double lo = w - hi;
    lo = lo + r;
    lo = __builtin_fma(kd, JK_LOG_LN2LO, lo);
    double r3 = r * r2;
    double q = __builtin_fma(r, A[4], A[3]);
    lo = __builtin_fma(r2, A[0], lo);
    q = __builtin_fma(q, r2, p1);
    double y = __builtin_fma(r3, q, lo);
    return y + hi;
}

// ---------------------------------------------------------------------------------------------
// libstdc++ std::gamma_distribution<double>(alpha, beta) for alpha >= 1 (Marsaglia-Tsang over the
// polar normal_distribution with its saved second deviate).  random.tcc:1802-1838, 2330-2392.
// State = {saved, saved_available}; it persists across reads of a lane exactly as the
// distribution object persists inside the reference's per-thread filler copy
// (src/hts_illumina.h:301,393).
// ---------------------------------------------------------------------------------------------
struct jk_gamma_state { double saved; int saved_available; int fail; };   // fail: a pow() argument left the transcribed main path
// a1 = malpha - 1/3, a2 = 1/sqrt(9*a1) with malpha = alpha (alpha >= 1) or alpha + 1 (alpha < 1: the draw is then
// multiplied by pow(u, 1/alpha), random.tcc:2336-2338,2380-2388)
struct jk_gamma_param { double a1; double a2; double beta; double inv_alpha; int small; };
JK_HD jk_gamma_param jk_gamma_make(double alpha, double beta) {
    jk_gamma_param p;
    const double malpha = alpha < 1.0 ? alpha + 1.0 : alpha;
    p.a1 = malpha - 1.0 / 3.0;
    p.a2 = 1.0 / __builtin_sqrt(9.0 * p.a1);
    p.beta = beta;
    p.small = alpha < 1.0;
    p.inv_alpha = 1.0 / alpha;
    return p;
}
JK_HD double jk_pow(double x, double y, bool* ok);      // jk_math2.h (glibc's pow, main path)

JK_HD double jk_sqrt(double v) { return __builtin_sqrt(v); }   // IEEE correctly rounded on both sides

template <typename Rng>
JK_HD double jk_normal(jk_gamma_state& st, Rng& rng) {
    if (st.saved_available) { st.saved_available = 0; return st.saved * 1.0 + 0.0; }
    double x, y, r2;
    do {
        x = 2.0 * jk_canonical(rng()) - 1.0;
        y = 2.0 * jk_canonical(rng()) - 1.0;
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    double mult = jk_sqrt(-2.0 * jk_log(r2) / r2);
    st.saved = x * mult;
    st.saved_available = 1;
    return (y * mult) * 1.0 + 0.0;
}

template <typename Rng>
JK_HD double jk_gamma(const jk_gamma_param& p, jk_gamma_state& st, Rng& rng) {
    double u, v, n;
    do {
        do {
            n = jk_normal(st, rng);
            v = 1.0 + p.a2 * n;
        } while (v <= 0.0);
        v = v * v * v;
        u = jk_canonical(rng());
    } while (u > 1.0 - 0.0331 * n * n * n * n &&
             (jk_log(u) > (0.5 * n * n + p.a1 * (1.0 - v + jk_log(v)))));
    if (!p.small) return p.a1 * v * p.beta;
    do u = jk_canonical(rng()); while (u == 0.0);
    bool ok = true;
    double pw = jk_pow(u, p.inv_alpha, &ok);
#if !defined(__HIP_DEVICE_COMPILE__)
    if (!ok) { pw = __builtin_pow(u, p.inv_alpha); ok = true; }      // host: libm itself (what the reference calls)
#endif
    if (!ok) st.fail = 1;
    return pw * p.a1 * v * p.beta;
}
