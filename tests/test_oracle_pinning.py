"""The oracle is only worth something if it is pinned to the reference.  These tests (CPU only)
check it against (1) the reference's own PCG headers compiled here (oracle/_ref) and the golden
vectors generated from them, and (2) every known-answer test the reference holds for the path
(tests/testthat/test-sequencer.R)."""
import json
import os

import numpy as np
import pytest

from helpers import job, run_oracle, fastq_records, write_test_profile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_pcg64_matches_golden_vectors_from_reference_headers(O):
    gold = json.load(open(os.path.join(HERE, "golden", "pcg64_vectors.json")))
    assert gold["max"] == "ffffffffffffffff"
    for st in gold["streams"]:
        out = O.pcg64_outputs(st["sub_seeds"], 64)
        assert ["%016x" % int(x) for x in out] == st["outputs"]


def test_pcg64_matches_reference_build_live(O):
    if O.ref_pcg_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference here); golden vectors cover it")
    rng = np.random.default_rng(7)
    for _ in range(50):
        w = rng.integers(0, 2 ** 32, size=8, dtype=np.uint64).astype(np.uint32)
        assert (O.pcg64_outputs(w, 256) == O.pcg64_outputs(w, 256, use_ref=True)).all()


@pytest.fixture(scope="module")
def ka(ja, tmp_path_factory):
    g = json.load(open(os.path.join(HERE, "golden", "sequencer_known_answers.json")))
    prof = write_test_profile(str(tmp_path_factory.mktemp("prof") / "test_prof.txt"))
    p = ja.read_profile(prof, None, g["read_length"], 1)
    genome = ja.RefGenome([g["chrom"]])
    return g, p, genome


@pytest.mark.parametrize("matepair", [False, True])
def test_reference_known_answer_pairs(O, ja, ka, matepair):
    """test-sequencer.R:91-161: C25 N150 T25 chromosome, fragments forced to 200, no indels ->
    every read is one of two strings; both must occur."""
    g, p, genome = ka
    words = ja.seed_words(99, 16 * 4)
    j = job(paired=True, matepair=matepair, frag_len_min=200, frag_len_max=200, ins_prob1=0, del_prob1=0,
            ins_prob2=0, del_prob2=0)
    r1, r2, _ = run_oracle(O, genome, p, p, words, g["n_reads"], 4, j)
    expect = g["matepair_expected_reads" if matepair else "paired_expected_reads"]
    for data in (r1, r2):
        recs = fastq_records(data)
        assert len(recs) == g["n_reads"] // 2
        reads = sorted(set(r[1].decode() for r in recs))
        assert reads == expect
        assert all(r[0].startswith(b"@REF-chrom0-") and r[2] == b"+" for r in recs)
        # quality 255 + '!' wraps to 32 (' ') for T/C/A/G; 'N' bases get '!'+0..9
        assert all(len(r[3]) == 100 for r in recs)


def test_reference_structural_checks(O, ja):
    """test-sequencer.R:31-77: 4 lines per read, '@' ids, '+' separators (SE and PE, 100 bp)."""
    genome = ja.synthetic_genome([100] * 5, seed=5)
    p1, p2 = ja.read_profile(None, None, 100, 1), ja.read_profile(None, None, 100, 2)
    words = ja.seed_words(3, 64)
    r1, r2, _ = run_oracle(O, genome, p1, None, words, 100, 1, job(paired=False))
    recs = fastq_records(r1)
    assert r2 is None and len(recs) == 100 and all(r[0][:1] == b"@" and r[2] == b"+" for r in recs)
    r1, r2, _ = run_oracle(O, genome, p1, p2, words, 100, 1, job(paired=True))
    for data in (r1, r2):
        recs = fastq_records(data)
        assert len(recs) == 50 and all(r[0][:1] == b"@" and r[2] == b"+" for r in recs)


def test_oracle_thread_window_and_discard(O, ja, hs25):
    genome = ja.synthetic_genome([50000], seed=2)
    words = ja.seed_words(1, 16 * 16)
    tb = {}
    j = job()
    a1, a2, used = run_oracle(O, genome, hs25[0], hs25[1], words, 2000, 16, j, thread_bytes=tb)
    assert used == 16 * 16
    b1, b2, _ = run_oracle(O, genome, hs25[0], hs25[1], words, 2000, 16, j, thread_begin=3, thread_end=5)
    off, n = int(tb[0][:3].sum()), int(tb[0][3:5].sum())
    assert a1[off:off + n] == b1
    c1, _, _ = run_oracle(O, genome, hs25[0], hs25[1], words, 2000, 16, j, discard=True)
    assert c1 == b""
