"""CPU-only: the arithmetic the kernels run (csrc/jk_math.h) evaluated on the host through the C ABI
and compared bit for bit with the oracle's x87 / libstdc++ / glibc expressions."""
import ctypes as C

import numpy as np
import pytest

from jackalope_amd import _abi


def host_eval(what, xs, aux=0):
    xs = np.ascontiguousarray(xs, dtype=np.uint64)
    stream = what in (_abi.OP_PCG_STREAM, _abi.OP_GAMMA_STREAM)
    n = xs.size // 8 if stream else xs.size
    out = np.zeros(n * aux if stream else n, dtype=np.uint64)
    _abi.check(_abi.lib().jk_host_eval(what, xs.ctypes.data, n, aux, out.ctypes.data))
    return out


def raw_inputs(n, seed=1):
    rng = np.random.default_rng(seed)
    xs = rng.integers(0, 2 ** 64, size=n, dtype=np.uint64)
    edge = [0, 1, 2, 2 ** 63 - 2, 2 ** 63 - 1, 2 ** 63, 2 ** 64 - 1, 2 ** 64 - 2, 2 ** 64 - 1024, 2 ** 64 - 1025,
            2 ** 64 - 2048, 2 ** 53, 2 ** 53 + 1, 2 ** 32, 2 ** 32 - 1]
    adv = []
    for n_ in [3, 5, 6, 7, 8, 10, 4, 33, 100, 255]:      # products just around k * 2^64
        for k in range(1, n_ + 1):
            t = (k << 64) // n_
            adv += [v for v in (t + d - 1 for d in range(-3, 4)) if 0 <= v < 2 ** 64]
    return np.concatenate([xs, np.array(edge, dtype=np.uint64), np.array(adv, dtype=np.uint64)])


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 10, 33, 41, 100, 255])
def test_runif_index(O, built, n):
    x = raw_inputs(300_000, seed=n)
    assert (host_eval(_abi.OP_RUNIF_INDEX, x, n) == O.eval_many(_abi.OP_RUNIF_INDEX, x, n)).all()


@pytest.mark.parametrize("what", [_abi.OP_RUNIF_DOUBLE, _abi.OP_CANONICAL, _abi.OP_N_QUAL, _abi.OP_LT_HALF])
def test_unary_conversions(O, built, what):
    x = raw_inputs(1_000_000, seed=what)
    assert (host_eval(what, x) == O.eval_many(what, x)).all()


@pytest.mark.parametrize("span", [1, 2, 3, 1000, 99_999_851, 2 ** 32, 3 * 10 ** 9])
def test_frag_start(O, built, span):
    x = raw_inputs(200_000, seed=span % 1000)
    assert (host_eval(_abi.OP_FRAG_START, x, span) == O.eval_many(_abi.OP_FRAG_START, x, span)).all()


def test_log_is_glibc_log(O, built):
    """jk_log restates glibc 2.35's log (FMA variant); std::gamma_distribution calls it."""
    rng = np.random.default_rng(11)
    n = 1_000_000
    d = np.concatenate([rng.random(n), 1 + (rng.random(n) - 0.5) * 0.13, rng.random(n) * 1e-300,
                        np.exp(rng.normal(0, 50, n)), np.array([1.0, 0.5, 2.0, 1e-310, 5e-324, np.inf, 0.0])])
    bits = d.astype(np.float64).view(np.uint64)
    a, b = host_eval(_abi.OP_LOG, bits), O.eval_many(_abi.OP_LOG, bits)
    assert (a == b).all(), "jk_log differs from this host's libm log on %d inputs" % int((a != b).sum())


def test_pcg_and_gamma_streams(O, built):
    rng = np.random.default_rng(5)
    sw = rng.integers(0, 2 ** 32, size=8 * 500, dtype=np.uint64)
    assert (host_eval(_abi.OP_PCG_STREAM, sw, 64) == O.eval_many(_abi.OP_PCG_STREAM, sw, 64)).all()
    for shape, scale in [(16.0, 25.0), (1.0, 300.0), (2.5, 7.0), (100.0, 4.0)]:
        _abi.lib().jk_eval_set_gamma(shape, scale)
        O.lib().orc_set_gamma(C.c_double(shape), C.c_double(scale))
        O.lib().orc_gamma_streams.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        g1 = host_eval(_abi.OP_GAMMA_STREAM, sw, 100)
        g2 = np.zeros(500 * 100, dtype=np.uint64)
        O.lib().orc_gamma_streams(sw.ctypes.data, 500, 100, g2.ctypes.data)
        assert (g1 == g2).all(), (shape, scale)
    _abi.lib().jk_eval_set_gamma(16.0, 25.0)
    O.lib().orc_set_gamma(C.c_double(16.0), C.c_double(25.0))


def test_alias_tables(O, built, hs25):
    L = _abi.lib()
    rng = np.random.default_rng(2)
    cases = [rng.random(k) for k in (1, 2, 3, 5, 8, 17, 40)] + [np.array([1.0]), np.array([0.5, 0.5]),
                                                                 np.array([1e-9, 1.0, 1e-9])]
    off = 0
    p = hs25[0]
    for k in p.n_quals.ravel()[:400]:            # real profile rows too
        cases.append(p.probs[off:off + k])
        off += k
    for probs in cases:
        probs = np.ascontiguousarray(probs, dtype=np.float64)
        P, A = np.zeros(probs.size), np.zeros(probs.size, dtype=np.uint64)
        L.jk_alias_build(probs.ctypes.data, probs.size, P.ctypes.data, A.ctypes.data)
        P2, A2 = O.alias_build(probs)
        assert (P.view(np.uint64) == P2.view(np.uint64)).all()
        # Alias[] is only meaningful where Prob < 1
        m = P < 1
        assert (A[m] == A2[m]).all()


def test_split_int_and_reads_per_group(O, built, ja):
    L = _abi.lib()
    for x, n in [(0, 1), (10, 3), (10_000_000, 1 << 20), (7, 7), (5, 9)]:
        a, b = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        L.jk_split_int(x, n, a.ctypes.data)
        O.lib().orc_split_int(C.c_uint64(x), C.c_uint64(n), b.ctypes.data_as(C.c_void_p))
        assert (a == b).all() and int(a.sum()) == x
    rng = np.random.default_rng(9)
    for n_reads in (0, 1, 7, 1000, 10 ** 6, 10 ** 9):
        for G in (1, 2, 5, 24):
            probs = rng.random(G) * 1e8
            if G > 2:
                probs[1] = 0.0
            words = ja.seed_words(n_reads + G, 64)
            out1, out2 = np.zeros(G, dtype=np.uint64), np.zeros(G, dtype=np.uint64)
            src = _abi.SeedSource()
            src.words = words.ctypes.data_as(C.POINTER(C.c_uint32))
            src.n_words = words.size
            _abi.check(L.jk_reads_per_group(n_reads, probs.ctypes.data, G, C.byref(src), out1.ctypes.data))
            used = C.c_uint64()
            rc = O.lib().orc_reads_per_group(C.c_uint64(n_reads), probs.ctypes.data_as(C.c_void_p), C.c_uint64(G),
                                             words.ctypes.data_as(C.c_void_p), C.c_uint64(words.size),
                                             out2.ctypes.data_as(C.c_void_p), C.byref(used))
            assert rc == 0 and (out1 == out2).all() and int(out1.sum()) == n_reads
            assert words.size - src.n_words == used.value == (8 if n_reads > 0 else 0)


def test_reads_per_group_fast_paths(O, built, ja):
    """The library precomputes reads_per_group's probability chain and restates libstdc++'s waiting-time binomial
    (t*p < 8) inline (csrc/jk_host.h: GroupChain, BinomDraw); the oracle calls std::binomial_distribution for every
    draw as the reference does (src/hts.h:58-103).  Random splits across both binomial branches, p > 0.5, zero and
    all-mass groups."""
    L = _abi.lib()
    rng = np.random.default_rng(2024)
    for case in range(4000):
        G = int(rng.integers(1, 31))
        kind = case % 5
        if kind == 0:
            probs = rng.random(G)
        elif kind == 1:
            probs = rng.random(G) ** 6 * 1e9                       # very uneven: p > 0.5 and tiny p
        elif kind == 2:
            probs = rng.random(G) * (rng.random(G) < 0.5)          # zeros
            if probs.sum() == 0:
                probs[-1] = 1.0
        elif kind == 3:
            probs = np.zeros(G)
            probs[int(rng.integers(0, G))] = 3.0                   # one group holds all the mass
        else:
            probs = np.full(G, 125e6) + rng.integers(-3, 4, size=G)   # chromosome sizes of configs[2]
        n_reads = int([rng.integers(1, 40), rng.integers(1, 400), rng.integers(1, 10 ** 5), rng.integers(1, 10 ** 10)][case % 4])
        words = ja.seed_words(case, 16)
        out1, out2 = np.zeros(G, dtype=np.uint64), np.zeros(G, dtype=np.uint64)
        src = _abi.SeedSource()
        src.words = words.ctypes.data_as(C.POINTER(C.c_uint32))
        src.n_words = words.size
        probs = np.ascontiguousarray(probs, dtype=np.float64)
        _abi.check(L.jk_reads_per_group(n_reads, probs.ctypes.data, G, C.byref(src), out1.ctypes.data))
        used = C.c_uint64()
        rc = O.lib().orc_reads_per_group(C.c_uint64(n_reads), probs.ctypes.data_as(C.c_void_p), C.c_uint64(G),
                                         words.ctypes.data_as(C.c_void_p), C.c_uint64(words.size),
                                         out2.ctypes.data_as(C.c_void_p), C.byref(used))
        assert rc == 0
        assert (out1 == out2).all(), (case, n_reads, probs, out1, out2)
        assert int(out1.sum()) == n_reads


# ---- PacBio arithmetic (csrc/jk_math2.h, jk_nmath.h) ---------------------------------------------

def eval2(fn, what, xs, per=1):
    xs = np.ascontiguousarray(xs, dtype=np.uint64)
    n = xs.size // per
    out = np.zeros(n, dtype=np.uint64)
    _abi.check(fn(what, xs.ctypes.data, n, 0, out.ctypes.data))
    return out


def orc2(O, what, xs, per=1):
    xs = np.ascontiguousarray(xs, dtype=np.uint64)
    n = xs.size // per
    out = np.zeros(n, dtype=np.uint64)
    O.lib().orc_eval_many(what, xs.ctypes.data, n, 0, out.ctypes.data)
    return out


def pacbio_math_inputs(n, seed):
    rng = np.random.default_rng(seed)
    d = {}
    d["exp"] = np.concatenate([rng.normal(0, 5, n), rng.uniform(-20, 20, n), rng.normal(9.8, 0.3, n),
                               rng.uniform(-1100, -500, n), rng.uniform(500, 1100, n),
                               np.array([0.0, 1e-20, -1e-20, 1.0, -1.0, -745.2, -744.9, 709.7, 709.9, -1e300, 1e300])])
    # (the last two blocks: results near and beyond the under- / overflow thresholds -- glibc's specialcase() -- as the
    #  gamma sampler's pow(u, 1 / shape) produces them for tiny shapes)
    bx = np.concatenate([rng.uniform(0.001, 1, n), np.full(n, 2.0), rng.uniform(1, 1e6, n), rng.uniform(0, 1, n) ** 4 + 1e-19,
                         rng.uniform(1.5, 40, n)])
    ey = np.concatenate([rng.uniform(0.6, 12, n), rng.uniform(-40, 10, n), np.full(n, 1.4691051212330266), rng.uniform(1, 2500, n),
                         rng.uniform(100, 1200, n)])
    xy = np.empty(2 * bx.size)
    xy[0::2], xy[1::2] = bx, ey
    d["pow"] = xy
    d["log10"] = np.concatenate([rng.uniform(1e-30, 1, n), rng.uniform(0, 1, n) ** 8, np.exp(rng.normal(0, 100, n))])
    d["qnorm"] = np.concatenate([rng.uniform(0, 1, n), rng.uniform(0, 1e-8, n), 1 - rng.uniform(0, 1e-8, n),
                                 rng.uniform(0, 1, n) ** 20])
    ps = np.concatenate([rng.uniform(0, 1, 200), rng.uniform(0, 1e-6, 100), 1 - rng.uniform(0, 1e-9, 100),
                         np.array([0.5, 0.25, 1e-300, 0.9999999999999999])])
    reps = max(n // ps.size, 1)
    xs = rng.integers(0, 2 ** 64, size=ps.size * reps, dtype=np.uint64)
    quad = np.zeros(4 * xs.size, dtype=np.uint64)
    quad[0::4] = xs
    quad[1::4] = np.repeat(ps, reps).view(np.uint64)
    for i, p in enumerate(ps):
        m, e = C.c_uint64(), C.c_int32()
        _abi.lib().jk_x87_one_minus(C.c_double(p), C.byref(m), C.byref(e))
        quad[4 * i * reps + 2:4 * (i + 1) * reps:4] = m.value
        quad[4 * i * reps + 3:4 * (i + 1) * reps:4] = np.int64(e.value).astype(np.uint64)
    d["runif_ab"] = quad
    return d


PB_OPS = [("exp", _abi.OP_EXP, 1), ("pow", _abi.OP_POW, 2), ("log10", _abi.OP_LOG10, 1), ("qnorm", _abi.OP_QNORM, 1),
          ("runif_ab", _abi.OP_RUNIF_AB, 4)]


@pytest.mark.parametrize("name,op,per", PB_OPS)
def test_pacbio_math_matches_host_libm_and_x87(O, built, name, op, per):
    """jk_exp / jk_pow / jk_log10 restate glibc 2.35 (FMA variants); runif_ab is x87 emulation; qnorm is AS 241."""
    x = pacbio_math_inputs(500_000, 3)[name]
    a = eval2(_abi.lib().jk_host_eval, op, x.view(np.uint64), per)
    b = orc2(O, op, x.view(np.uint64), per)
    ok = a != np.uint64(2 ** 64 - 1)          # ~0 marks "outside the transcribed paths"
    assert ok.mean() > 0.99 and (ok.all() or name not in ("exp", "pow"))
    assert (a[ok] == b[ok]).all(), "%s differs on %d inputs" % (name, int((a[ok] != b[ok]).sum()))


def test_nmath_restatements_against_scipy(O, built):
    """pnorm (Cody) and qchisq are restatements of published algorithms (R itself is unavailable: parity with R
    unpinned); they must at least agree with an independent implementation to rounding level."""
    from scipy.stats import norm, chi2
    from scipy.special import ndtri
    O.lib().orc_pnorm.restype = C.c_double
    O.lib().orc_pnorm.argtypes = [C.c_double]
    O.lib().orc_qchisq.restype = C.c_double
    O.lib().orc_qchisq.argtypes = [C.c_double, C.c_double]
    for x in np.linspace(-30, 8, 400):
        assert abs(O.lib().orc_pnorm(x) - norm.cdf(x)) <= 5e-13 * norm.cdf(x) + 1e-300
    for df in list(np.linspace(2.0, 13, 60)) + [0.5, 50, 300]:
        assert abs(O.lib().orc_qchisq(0.9925, df) - chi2.ppf(0.9925, df)) <= 1e-12 * chi2.ppf(0.9925, df)
    p = np.random.default_rng(1).uniform(0, 1, 20000)
    q = eval2(_abi.lib().jk_host_eval, _abi.OP_QNORM, p.view(np.uint64)).view(np.float64)
    assert np.max(np.abs(q - ndtri(p)) / np.maximum(np.abs(ndtri(p)), 1e-12)) < 1e-13


def test_nmath_published_check_values_and_exact_references(O, built):
    """What can be pinned without R.  (1) Wichura's AS 241 paper gives check values for PPND16: the restated qnorm must
    return them.  (2) pnorm (Cody 1969) and qchisq -- R's own are accurate to about 1e-15 -- are compared with 50-digit
    references (mpmath) over the ranges the PacBio set-up uses them in: the thresholds qchisq(0.9925, n) for n from the
    clamp 0.001 up to chi2_params_n's cap, pnorm over the truncated-normal bounds.  A 1-ulp difference from R in one of
    these table values changes a read only when a draw falls into that ulp (probability ~1e-16 per draw)."""
    import mpmath as mp
    mp.mp.dps = 50
    # (1) AS 241, "Test data": PPND16(0.25), PPND16(0.001), PPND16(1e-20)
    as241 = [(0.25, -0.6744897501960817), (0.001, -3.090232306167814), (1e-20, -9.262340089798408)]
    ps = np.array([p for p, _ in as241])
    got_host = eval2(_abi.lib().jk_host_eval, _abi.OP_QNORM, ps.view(np.uint64)).view(np.float64)
    got_orc = orc2(O, _abi.OP_QNORM, ps.view(np.uint64)).view(np.float64)
    for (p, want), a, b in zip(as241, got_host, got_orc):
        assert a == b
        assert abs(a - want) <= 5e-16 * abs(want), (p, a, want)       # (the paper prints 16 significant digits)
    # by symmetry and at the branch points of the algorithm (|q| = 0.425, r = 5)
    for p in (0.5, 0.075, 0.925, float(mp.exp(-25)), 1 - 2.0 ** -30):
        want = float(mp.sqrt(2) * mp.erfinv(2 * mp.mpf(p) - 1))
        a = eval2(_abi.lib().jk_host_eval, _abi.OP_QNORM, np.array([p]).view(np.uint64)).view(np.float64)[0]
        assert abs(a - want) <= 4e-16 * max(abs(want), 1e-300) + (1e-16 if p == 0.5 else 0), (p, a, want)
    # (2) pnorm against exact values, lower tail to the underflow edge
    O.lib().orc_pnorm.restype = C.c_double
    O.lib().orc_pnorm.argtypes = [C.c_double]
    worst = 0.0
    for x in list(np.linspace(-37.5, 8.2, 300)) + [0.0, -0.67448, 0.66291, -5.656854, 5.656854, -1e-8]:
        want = mp.ncdf(mp.mpf(float(x)))
        got = O.lib().orc_pnorm(float(x))
        rel = abs((mp.mpf(got) - want) / want)
        worst = max(worst, float(rel))
    assert worst < 8e-16, worst                 # Cody's rational approximations: a few ulp at most
    O.lib().orc_qchisq.restype = C.c_double
    O.lib().orc_qchisq.argtypes = [C.c_double, C.c_double]
    worst = 0.0
    for df in [0.001, 0.01, 0.1, 0.5, 1.0, 1.5, 2.0] + list(np.linspace(2.5, 13.0, 43)) + [25.0, 100.0]:
        half = mp.mpf(df) / 2
        f = lambda x: mp.gammainc(half, 0, x / 2, regularized=True) - mp.mpf("0.9925")
        guess = O.lib().orc_qchisq(0.9925, float(df))
        want = mp.findroot(f, mp.mpf(guess))
        worst = max(worst, float(abs((mp.mpf(guess) - want) / want)))
    # measured: 9.8e-14 (the Newton solve stops at 1e-13; R's own qgamma stops its single Newton polish at |dp| < 1e-15 p from
    # an AS 91 start good to 5e-7, i.e. it is not exact to the ulp either): agreement with R to ~1e-13 is the honest claim
    assert worst < 2e-13, worst
