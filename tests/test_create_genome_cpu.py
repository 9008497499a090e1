"""create_genome, CPU side: the LCG jump-ahead that lets the device fill all bases in parallel
(jk_pcg_advance_outputs) against the reference engine's own advance() built from
/root/reference/inst/include/pcg (oracle/_ref/libref_pcg.so) and against plain stepping, and the
oracle's restatement of create_chromosomes_ (src/create_sequences.cpp:59-138) against the reference's
structural test (tests/testthat/test-R_classes.R:15-31)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as OL
from jackalope_amd import _abi
from jackalope_amd.rng import seed_words


def jumped(words8, steps, n):
    w = np.ascontiguousarray(words8, dtype=np.uint32)
    out = np.zeros(n, dtype=np.uint64)
    _abi.lib().jk_pcg_advance_outputs(w.ctypes.data, steps, n, out.ctypes.data)
    return out


@pytest.mark.parametrize("steps", [0, 1, 2, 3, 255, 4096, 2 * 2048 * 12345, (1 << 40) + 12345, (1 << 63) + 977, (1 << 64) - 1])
def test_jump_matches_reference_advance(built, steps):
    ref = OL.ref_pcg_lib()
    if ref is None:
        pytest.skip("oracle/_ref not built (no reference tree here)")
    for row in range(3):
        w = seed_words(100 + row, 8)
        want = np.zeros(16, dtype=np.uint64)
        ref.ref_pcg64_advance_outputs(w.ctypes.data_as(C.c_void_p), C.c_uint64(0), C.c_uint64(steps), C.c_uint64(16),
                                      want.ctypes.data_as(C.c_void_p))
        assert np.array_equal(jumped(w, steps, 16), want)


def test_jump_matches_stepping(built):
    w = seed_words(7, 8)
    seq = OL.pcg64_outputs(w, 5000)
    for steps in (0, 1, 2, 17, 1000, 4095, 4096, 4983):
        assert np.array_equal(jumped(w, steps, 16), seq[steps:steps + 16])


def test_oracle_create_genome_structure(O):
    """rando_chroms(10, 100, 10, pi_tcag = c(8, 4, 2, 1)): observed base frequencies rank like pi_tcag."""
    chroms, used = O.create_genome(10, 100.0, 10.0, [8, 4, 2, 1], 1, seed_words(3, 8))
    assert used == 8 and len(chroms) == 10
    joined = b"".join(chroms)
    assert set(joined) <= set(b"TCAG")
    freq = [joined.count(c) / len(joined) for c in b"TCAG"]
    assert sorted(freq, reverse=True) == freq
    lens = np.array([len(c) for c in chroms])
    assert 70 < lens.mean() < 130 and lens.std() > 0
    same, _ = O.create_genome(3, 50.0, 0.0, [1, 1, 1, 1], 1, seed_words(3, 8))
    assert [len(c) for c in same] == [50, 50, 50]


def test_oracle_thread_blocks(O):
    """omp for schedule(static): thread t's chromosomes depend only on thread t's seed row."""
    words = seed_words(11, 8 * 3)
    all3, used = O.create_genome(8, 200.0, 20.0, [1, 2, 3, 4], 3, words)       # blocks of 3, 3, 2
    assert used == 24
    first, _ = O.create_genome(3, 200.0, 20.0, [1, 2, 3, 4], 1, words[:8])
    second, _ = O.create_genome(3, 200.0, 20.0, [1, 2, 3, 4], 1, words[8:16])
    third, _ = O.create_genome(2, 200.0, 20.0, [1, 2, 3, 4], 1, words[16:24])
    assert all3 == first + second + third
