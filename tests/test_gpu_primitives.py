"""GPU: every arithmetic primitive of the path evaluated ON THE DEVICE (jk_dev_eval runs the same
inline functions the kernels use, one thread per element) against the oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from jackalope_amd import _abi
from test_host_primitives import raw_inputs

pytestmark = pytest.mark.gpu


def dev_eval(what, xs, aux=0):
    xs = np.ascontiguousarray(xs, dtype=np.uint64)
    stream = what in (_abi.OP_PCG_STREAM, _abi.OP_GAMMA_STREAM)
    n = xs.size // 8 if stream else xs.size
    out = np.zeros(n * aux if stream else n, dtype=np.uint64)
    _abi.check(_abi.lib().jk_dev_eval(0, what, xs.ctypes.data, n, aux, out.ctypes.data))
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8, 10, 41, 255])
def test_runif_index(O, built, n):
    x = raw_inputs(1_000_000, seed=100 + n)
    assert (dev_eval(_abi.OP_RUNIF_INDEX, x, n) == O.eval_many(_abi.OP_RUNIF_INDEX, x, n)).all()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 10, 33, 41, 100, 255])
def test_runif_index_32bit_kernel_form(O, built, n):
    """runif_index32 (what the per-base loops call), including inputs whose product lands within a few units of a
    multiple of 2^64 -- the only place where the x87 rounding can carry (a wave-uniform rare branch in the kernel)."""
    x = raw_inputs(500_000, seed=300 + n)
    k = np.arange(1, n + 1, dtype=object)
    near = np.array([int(v) for kk in k for v in (((int(kk) << 64) + d) // n - 1 for d in range(-2 * n, 2 * n + 1)) if 0 <= v < 2 ** 64],
                    dtype=np.uint64)
    x = np.concatenate([x, near])
    assert (dev_eval(_abi.OP_RUNIF_INDEX32, x, n) == O.eval_many(_abi.OP_RUNIF_INDEX, x, n)).all()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 37, 40, 128, 254, 255])
def test_alias_index_of_the_quality_step(O, built, n):
    """alias_index32: mul_hi(x_hi, n), with the 96-bit routine behind a filter on the low word of x_hi*n -- inputs whose
    x_hi*n lands within a few hundred of a multiple of 2^32 (where the filter fires or just does not), with random low
    words and the extreme ones."""
    rng = np.random.default_rng(900 + n)
    xh = np.array([v for k in range(1, n + 1) for v in (((k << 32) + d) // n for d in range(-300, 301)) if 0 <= v < 2 ** 32], dtype=np.uint64)
    lows = np.concatenate([rng.integers(0, 2 ** 32, size=xh.size, dtype=np.uint64), np.zeros(xh.size, np.uint64),
                           np.full(xh.size, 2 ** 32 - 1, np.uint64), np.full(xh.size, 2 ** 32 - 2, np.uint64)])
    x = np.concatenate([(np.tile(xh, 4) << np.uint64(32)) | lows, raw_inputs(200_000, seed=500 + n)])
    assert (dev_eval(_abi.OP_ALIAS_INDEX32, x, n) == O.eval_many(_abi.OP_RUNIF_INDEX, x, n)).all()


def test_n_qual_kernel_form_near_integer_boundaries(O, built):
    """n_qual32 (what the kernels call for a non-TCAG base): inputs whose (x+1)*10 lies within a few hundred units of a
    multiple of 2^64 -- the only place where the two x87 roundings can change the integer part."""
    near = np.array([v for k in range(1, 11) for v in (((k << 64) + d) // 10 - 1 for d in range(-400, 401)) if 0 <= v < 2 ** 64],
                    dtype=np.uint64)
    x = np.concatenate([raw_inputs(200_000, seed=77), near, np.array([0, 1, 2 ** 64 - 1, 2 ** 64 - 2, 2 ** 63], dtype=np.uint64)])
    assert (dev_eval(_abi.OP_N_QUAL, x) == O.eval_many(_abi.OP_N_QUAL, x)).all()


@pytest.mark.parametrize("what", [_abi.OP_RUNIF_DOUBLE, _abi.OP_CANONICAL, _abi.OP_N_QUAL, _abi.OP_LT_HALF])
def test_unary_conversions(O, built, what):
    x = raw_inputs(4_000_000, seed=200 + what)
    assert (dev_eval(what, x) == O.eval_many(what, x)).all()


@pytest.mark.parametrize("span", [1, 3, 1000, 99_999_851, 2 ** 32, 3 * 10 ** 9])
def test_frag_start(O, built, span):
    x = raw_inputs(1_000_000, seed=span % 1000)
    assert (dev_eval(_abi.OP_FRAG_START, x, span) == O.eval_many(_abi.OP_FRAG_START, x, span)).all()


def test_log_and_sqrt_match_host_libm(O, built):
    rng = np.random.default_rng(12)
    n = 4_000_000
    d = np.concatenate([rng.random(n), 1 + (rng.random(n) - 0.5) * 0.13, rng.random(n) * 1e-300,
                        np.exp(rng.normal(0, 50, n)), np.array([1.0, 0.5, 2.0, 1e-310, 5e-324, np.inf])])
    bits = d.astype(np.float64).view(np.uint64)
    assert (dev_eval(_abi.OP_LOG, bits) == O.eval_many(_abi.OP_LOG, bits)).all()
    assert (dev_eval(_abi.OP_SQRT, bits) == O.eval_many(_abi.OP_SQRT, bits)).all()


def test_pcg_and_gamma_streams(O, built):
    rng = np.random.default_rng(6)
    sw = rng.integers(0, 2 ** 32, size=8 * 4096, dtype=np.uint64)
    assert (dev_eval(_abi.OP_PCG_STREAM, sw, 128) == O.eval_many(_abi.OP_PCG_STREAM, sw, 128)).all()
    O.lib().orc_gamma_streams.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    for shape, scale in [(16.0, 25.0), (1.0, 300.0), (2.5, 7.0)]:
        _abi.lib().jk_eval_set_gamma(shape, scale)
        O.lib().orc_set_gamma(C.c_double(shape), C.c_double(scale))
        g1 = dev_eval(_abi.OP_GAMMA_STREAM, sw, 200)
        g2 = np.zeros(4096 * 200, dtype=np.uint64)
        O.lib().orc_gamma_streams(sw.ctypes.data, 4096, 200, g2.ctypes.data)
        assert (g1 == g2).all(), (shape, scale)
    _abi.lib().jk_eval_set_gamma(16.0, 25.0)
    O.lib().orc_set_gamma(C.c_double(16.0), C.c_double(25.0))


@pytest.mark.parametrize("name,op,per", __import__("test_host_primitives").PB_OPS)
def test_pacbio_math_on_device(O, built, name, op, per):
    from test_host_primitives import pacbio_math_inputs, orc2
    x = pacbio_math_inputs(2_000_000, 4)[name].view(np.uint64)
    n = x.size // per
    out = np.zeros(n, dtype=np.uint64)
    _abi.check(_abi.lib().jk_dev_eval(0, op, np.ascontiguousarray(x).ctypes.data, n, 0, out.ctypes.data))
    b = orc2(O, op, x, per)
    ok = out != np.uint64(2 ** 64 - 1)
    assert ok.mean() > 0.99 and (out[ok] == b[ok]).all(), "%s differs on device" % name
