"""GPU parity for create_genome (create_genome_cpp, src/create_sequences.cpp:59-169): the device fills all
bases in parallel through LCG jump-ahead; the oracle walks every engine sequentially as the reference
does.  Chromosome lengths and bases must be identical."""
import numpy as np
import pytest

from helpers import job, first_diff

pytestmark = pytest.mark.gpu


def check(ja, O, n_chroms, len_mean, len_sd, pi, T, seed):
    words = ja.seed_words(seed, 8 * T)
    want, used = O.create_genome(n_chroms, len_mean, len_sd, pi, T, words)
    g = ja.create_genome(n_chroms, len_mean, len_sd, pi, T, seed_words=words)
    assert g.seed_words_used() == used
    assert g.sizes() == [len(c) for c in want]
    assert g.names == ["chrom%d" % i for i in range(n_chroms)]
    for i in range(n_chroms):
        got = g.chrom(i).tobytes()
        if got != want[i]:
            raise AssertionError("chromosome %d differs at byte %d: %r vs %r" % ((i,) + first_diff(got, want[i])))
    return g


def test_reference_test_shape(ja, O):
    """tests/testthat/test-R_classes.R:15: 10 chromosomes, mean 100, sd 10, pi_tcag = 8:4:2:1"""
    g = check(ja, O, 10, 100, 10.0, [8, 4, 2, 1], 1, seed=1)
    joined = b"".join(s.tobytes() for s in g.seqs)
    freq = [joined.count(c) for c in b"TCAG"]
    assert sorted(freq, reverse=True) == freq


@pytest.mark.parametrize("n_chroms,len_mean,len_sd,T", [
    (1, 1, 0, 1), (1, 2047, 0, 1), (1, 2048, 0, 1), (1, 2049, 0, 1), (3, 100_000, 0, 1),
    (24, 50_000, 20_000.0, 1), (24, 50_000, 20_000.0, 5), (7, 300_000, 1000.0, 64), (200, 10.0, 9.0, 3),
])
def test_lengths_runs_and_threads(ja, O, n_chroms, len_mean, len_sd, T):
    check(ja, O, n_chroms, len_mean, len_sd, [0.1, 0.4, 0.3, 0.2], T, seed=n_chroms + T)


def test_len_sd_above_len_mean(ja, O):
    """gamma shape (mean / sd)^2 < 1 (src/create_sequences.cpp:85-95 puts no bound on len_sd): libstdc++'s pow branch."""
    check(ja, O, 40, 3000.0, 4500.0, [0.25] * 4, 3, seed=77)
    check(ja, O, 25, 500.0, 2000.0, [0.3, 0.2, 0.2, 0.3], 2, seed=78)


def test_degenerate_frequencies(ja, O):
    check(ja, O, 2, 5000, 0, [0, 0, 1, 0], 1, seed=5)            # only A
    check(ja, O, 2, 5000, 0, [1, 0, 0, 1e-9], 1, seed=6)
    check(ja, O, 2, 5000, 0, [0.25, 0.25, 0.25, 0.25], 1, seed=7)


def test_made_genome_feeds_the_sequencer_in_place(ja, O):
    """illumina() on the device-resident genome == illumina() on its host copy == oracle."""
    words_g = ja.seed_words(9, 8)
    g = ja.create_genome(3, 40_000, 5000.0, [0.3, 0.2, 0.3, 0.2], 1, seed_words=words_g)
    host = ja.RefGenome([s.copy() for s in g.seqs], names=g.names)
    T, n = 16, 4000
    words = ja.seed_words(10, 16 * T)
    with ja.illumina(g, None, n, 150, True, n_threads=T, seed_words=words, _session=True) as s:
        s.generate()
        d1, d2 = s.fetch(0), s.fetch(1)
    with ja.illumina(host, None, n, 150, True, n_threads=T, seed_words=words, _session=True) as s:
        s.generate()
        assert (s.fetch(0), s.fetch(1)) == (d1, d2)
    p1, p2 = ja.read_profile(None, None, 150, 1), ja.read_profile(None, None, 150, 2)
    j = job()
    o1, o2, _ = O.illumina_ref(host, paired=True, n_reads=n, prob_dup=j["prob_dup"], n_threads=T, read_pool_size=1000,
                               shape=16.0, scale=25.0, fmin=150, fmax=2 ** 32 - 1, prof1=p1, prof2=p2, ins1=j["ins_prob1"],
                               del1=j["del_prob1"], ins2=j["ins_prob2"], del2=j["del_prob2"], words=words)
    assert (d1, d2) == (o1, o2)


def test_argument_errors(ja):
    with pytest.raises(ValueError, match="argument `n_chroms` must be a single integer >= 1"):
        ja.create_genome(0, 100)
    with pytest.raises(ValueError, match="argument `len_mean`"):
        ja.create_genome(1, 0.5)
    with pytest.raises(ValueError, match="argument `pi_tcag`"):
        ja.create_genome(1, 100, pi_tcag=[0, 0, 0, 0])
    with pytest.raises(ja.JackalopeHipError, match="seed"):
        ja.create_genome(4, 100, n_threads=2, seed_words=ja.seed_words(1, 8))
