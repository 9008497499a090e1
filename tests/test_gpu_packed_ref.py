"""GPU: the 2-bit copy of the reference (GenomeDev::packed, csrc/jk_illumina_kernel.h) against the byte path and the
oracle.  A read end takes all its bases from the packed copy unless its source window touches a 4096-base block in which
a chromosome holds a byte other than T, C, A, G; JK_PACKED_REF=0 switches the copy off.  Every case runs both ways and
is compared with the oracle byte for byte (src/ref_classes.h:38-39 stores one char per base: the packing is this path's
own).  Chromosomes here are long enough for unflagged blocks to exist next to flagged ones."""
import numpy as np
import pytest

from helpers import job
from test_gpu_parity import check

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["packed", "bytes"], autouse=True)
def ref_mode(request, monkeypatch):
    monkeypatch.setenv("JK_PACKED_REF", "1" if request.param == "packed" else "0")
    return request.param


def genome_with_n_runs(ja, n, seed, runs):
    rng = np.random.default_rng(seed)
    seq = np.frombuffer(b"TCAG", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
    for start, length in runs:
        seq[start:start + length] = ord("N")
    return seq


def test_plain_genome_all_alignments(ja, O):
    # chromosome lengths that are not multiples of 4 or 64; many lanes so that every (start & 3, strand) occurs
    g = ja.synthetic_genome([100_003, 77_777, 12_345, 1_001], seed=41)
    check(ja, O, g, 150, 20_000, 257, job())
    check(ja, O, g, 150, 6_000, 64, job(matepair=True))
    check(ja, O, g, 100, 5_000, 64, job(paired=False))


def test_n_runs_mixed_waves(ja, O):
    # isolated N's, short runs and long runs: lanes of one wave on both paths at once, windows that end right before /
    # start right after a flagged block
    runs = [(1_000, 1), (5_000, 3), (9_984, 64), (20_000, 700), (40_001, 63), (60_000, 5_000), (90_000, 2)]
    seq = genome_with_n_runs(ja, 120_000, 42, runs)
    g = ja.RefGenome([seq, genome_with_n_runs(ja, 30_000, 43, [(0, 200), (29_800, 200)])])
    check(ja, O, g, 150, 16_000, 200, job())
    check(ja, O, g, 150, 8_000, 64, job(frag_mean=200.0, frag_sd=30.0))


def test_indels_barcodes_short_fragments(ja, O):
    g = ja.synthetic_genome([50_000, 311], seed=44)
    check(ja, O, g, 150, 6_000, 70, job(ins_prob1=0.02, del_prob1=0.03, ins_prob2=0.03, del_prob2=0.02))
    check(ja, O, g, 150, 4_000, 64, job(barcode="ACGTTGCA"))
    check(ja, O, g, 150, 4_000, 64, job(frag_mean=160.0, frag_sd=60.0, frag_len_min=20))


def test_other_read_lengths(ja, O):
    g = ja.synthetic_genome([90_001], seed=45)
    for L in (36, 100, 250):
        check(ja, O, g, L, 3_000, 64, job(frag_mean=600.0, frag_sd=100.0))
