// jk_nmath.h -- host-only restatements of the pieces of R's nmath the PacBio set-up needs
// (R::pnorm5 and R::qchisq; reference call sites src/hts_pacbio.h:178,349).  R is not part of
// /root/reference and is not installed here, so these follow the published algorithms R implements:
//   pnorm : W. J. Cody (1969), "Rational Chebyshev approximations for the error function" (the
//           three-range scheme with the exp(-xsq^2/2) exp(-del/2) split);
//   qchisq: quantile of the chi-square distribution by bracketed Newton iterations on the regularised
//           incomplete gamma function (series / Lentz continued fraction, long double).
// Parity with R itself is UNPINNED; tests compare them with scipy to ~1e-13 relative.  They only feed
// per-run tables (one entry per pass count / read length), never the per-base path.
#pragma once
#include <cfloat>
#include <cmath>

namespace jk {

inline double pnorm_std(double x) {          // P[N(0,1) <= x]
    static const double a[5] = {2.2352520354606839287, 161.02823106855587881, 1067.6894854603709582,
                                18154.981253343561249, 0.065682337918207449113};
    static const double b[4] = {47.20258190468824187, 976.09855173777669322, 10260.932208618978205,
                                45507.789335026729956};
    static const double c[9] = {0.39894151208813466764, 8.8831497943883759412, 93.506656132177855979,
                                597.27027639480026226, 2494.5375852903726711, 6848.1904505362823326,
                                11602.651437647350124, 9842.7148383839780218, 1.0765576773720192317e-8};
    static const double d[8] = {22.266688044328115691, 235.38790178262499861, 1519.377599407554805,
                                6485.558298266760755, 18615.571640885098091, 34900.952721145977266,
                                38912.003286093271411, 19685.429676859990727};
    static const double p[6] = {0.21589853405795699, 0.1274011611602473639, 0.022235277870649807,
                                0.001421619193227893466, 2.9112874951168792e-5, 0.02307344176494017303};
    static const double q[5] = {1.28426009614491121, 0.468238212480865118, 0.0659881378689285515,
                                0.00378239633202758244, 7.29751555083966205e-5};
    const double eps = DBL_EPSILON * 0.5, y = std::fabs(x);
    double xden, xnum, temp, del, xsq, cum, ccum;
    if (std::isnan(x)) return x;
    if (y <= 0.67448975) {
        if (y > eps) {
            xsq = x * x; xnum = a[4] * xsq; xden = xsq;
            for (int i = 0; i < 3; ++i) { xnum = (xnum + a[i]) * xsq; xden = (xden + b[i]) * xsq; }
        } else xnum = xden = 0.0;
        temp = x * (xnum + a[3]) / (xden + b[3]);
        return 0.5 + temp;
    }
    if (y <= 5.656854249492380195206754896838 /* sqrt(32) */) {
        xnum = c[8] * y; xden = y;
        for (int i = 0; i < 7; ++i) { xnum = (xnum + c[i]) * y; xden = (xden + d[i]) * y; }
        temp = (xnum + c[7]) / (xden + d[7]);
        xsq = std::trunc(y * 16) / 16; del = (y - xsq) * (y + xsq);
        cum = std::exp(-xsq * xsq * 0.5) * std::exp(-del * 0.5) * temp; ccum = 1.0 - cum;
        return x > 0. ? ccum : cum;
    }
    if (x > -37.5193 && x < 8.2924) {
        xsq = 1.0 / (x * x);
        xnum = p[5] * xsq; xden = xsq;
        for (int i = 0; i < 4; ++i) { xnum = (xnum + p[i]) * xsq; xden = (xden + q[i]) * xsq; }
        temp = xsq * (xnum + p[4]) / (xden + q[4]);
        temp = (0.398942280401432677939946059934 /* 1/sqrt(2 pi) */ - temp) / y;
        xsq = std::trunc(x * 16) / 16; del = (x - xsq) * (x + xsq);
        cum = std::exp(-xsq * xsq * 0.5) * std::exp(-del * 0.5) * temp; ccum = 1.0 - cum;
        return x > 0. ? ccum : cum;
    }
    return x > 0 ? 1.0 : 0.0;
}

// regularised lower incomplete gamma P(a, x), a > 0, x >= 0
inline long double gamma_p(long double a, long double x) {
    if (x <= 0) return 0;
    const long double lg = lgammal(a);
    if (x < a + 1) {                                   // series
        long double ap = a, del = 1 / a, sum = del;
        for (int n = 0; n < 100000; n++) {
            ap += 1; del *= x / ap; sum += del;
            if (fabsl(del) < fabsl(sum) * 1e-20L) break;
        }
        return sum * expl(-x + a * logl(x) - lg);
    }
    const long double tiny = 1e-4000L;                 // Lentz continued fraction for Q(a, x)
    long double bb = x + 1 - a, cc = 1 / tiny, dd = 1 / bb, h = dd;
    for (int i = 1; i < 100000; i++) {
        const long double an = -i * (i - a);
        bb += 2;
        dd = an * dd + bb; if (fabsl(dd) < tiny) dd = tiny;
        cc = bb + an / cc; if (fabsl(cc) < tiny) cc = tiny;
        dd = 1 / dd;
        const long double dl = dd * cc;
        h *= dl;
        if (fabsl(dl - 1) < 1e-20L) break;
    }
    return 1 - expl(-x + a * logl(x) - lg) * h;
}

inline double qchisq_upper_tail_point(double p, double df) {      // x with P[chi2_df <= x] = p
    const long double a = 0.5L * df;
    // Wilson-Hilferty start, then safeguarded Newton on P(a, x/2) - p
    long double lo = 0, hi = 1;
    while (gamma_p(a, hi / 2) < p) { lo = hi; hi *= 2; if (hi > 1e300L) return INFINITY; }
    long double x = 0.5L * (lo + hi);
    for (int it = 0; it < 200; it++) {
        const long double f = gamma_p(a, x / 2) - p;
        if (f > 0) hi = x; else lo = x;
        // density of chi2_df at x
        const long double dens = 0.5L * expl((a - 1) * logl(x / 2) - x / 2 - lgammal(a));
        long double xn = x - f / dens;
        if (!(xn > lo && xn < hi)) xn = 0.5L * (lo + hi);
        if (fabsl(xn - x) <= 1e-18L * fabsl(x)) { x = xn; break; }
        x = xn;
    }
    return static_cast<double>(x);
}

}  // namespace jk
