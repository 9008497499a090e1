"""GPU parity tests proper: the HIP path, called through the C ABI (via the illumina() mirror),
against the CPU oracle on the same seeded inputs -- FASTQ bytes must be identical.  The cases follow
the reference's own sequencer tests (tests/testthat/test-sequencer.R) plus the edge cases of the
path: single-end / paired / mate-pair, barcodes, duplicates with pool boundaries, short fragments,
N runs, many chromosomes, indel-heavy error models, empty lanes, profiles too big for LDS."""
import json
import os

import numpy as np
import pytest

from helpers import job, run_oracle, run_hip, open_hip, fastq_records, write_test_profile, first_diff

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def check(ja, O, genome, read_length, n_reads, n_threads, j, seed=1, profiles=(None, None), seq_sys=None):
    p1f, p2f = profiles
    paired = j["paired"] or j["matepair"]
    p1 = ja.read_profile(p1f, None, read_length, 1)
    p2 = ja.read_profile(p2f, None, read_length, 2) if paired else None
    words = ja.seed_words(seed, 16 * n_threads + 64)
    o1, o2, used_o = run_oracle(O, genome, p1, p2, words, n_reads, n_threads, j)
    h1, h2, reads, used_h = run_hip(ja, genome, profiles, read_length, words, n_reads, n_threads, j)
    assert used_o == used_h
    if h1 != o1:
        raise AssertionError("R1 differs at byte %d:\nHIP    %r\noracle %r" % first_diff(h1, o1))
    if paired and h2 != o2:
        raise AssertionError("R2 differs at byte %d:\nHIP    %r\noracle %r" % first_diff(h2, o2))
    ends = 2 if paired else 1
    assert reads == (n_reads // ends) * ends
    return h1, h2


def test_config1_plumbing_case(ja, O):
    """BASELINE configs[0]: 10 kb genome, 1k PE150 reads, n_threads = 1."""
    g = ja.synthetic_genome([10_000], seed=1)
    h1, h2 = check(ja, O, g, 150, 1000, 1, job())
    assert len(fastq_records(h1)) == 500 and len(fastq_records(h2)) == 500


@pytest.mark.parametrize("n_threads", [2, 7, 64, 1000, 5000])
def test_lane_counts_including_empty_lanes(ja, O, n_threads):
    g = ja.synthetic_genome([200_000], seed=2)
    check(ja, O, g, 150, 6000, n_threads, job(), seed=n_threads)


def test_single_end(ja, O):
    g = ja.synthetic_genome([50_000], seed=3)
    check(ja, O, g, 150, 3001, 33, job(paired=False))


def test_matepair(ja, O):
    g = ja.synthetic_genome([50_000], seed=4)
    check(ja, O, g, 150, 3000, 17, job(matepair=True, frag_mean=1000.0, frag_sd=150.0))


@pytest.mark.parametrize("barcode", ["ACGT", "TTGACCAN", "A" * 32, "GATTACA" * 9, "TCAG" * 37])      # the last: 148 of 150 bases
def test_barcodes(ja, O, barcode):
    g = ja.synthetic_genome([50_000], seed=5)
    check(ja, O, g, 150, 2000, 9, job(barcode=barcode))
    check(ja, O, g, 150, 1000, 5, job(barcode=barcode, paired=False))


@pytest.mark.parametrize("prob_dup,pool", [(0.0, 1000), (0.5, 1000), (1.0, 1000), (0.5, 1), (0.9, 2), (0.9, 6), (0.3, 7)])
def test_duplicates_and_pool_boundaries(ja, O, prob_dup, pool):
    g = ja.synthetic_genome([80_000], seed=6)
    check(ja, O, g, 150, 4000, 13, job(prob_dup=prob_dup, read_pool_size=pool), seed=pool)


def test_short_fragments_and_tiny_chromosomes(ja, O):
    g = ja.synthetic_genome([60_000], seed=7)
    check(ja, O, g, 150, 2000, 11, job(frag_len_min=40, frag_len_max=120))          # reads shorter than read_length
    check(ja, O, g, 150, 2000, 11, job(frag_len_min=1, frag_len_max=10, ins_prob1=0.05, del_prob1=0.05))
    tiny = ja.synthetic_genome([90, 149, 150, 151, 300, 35], seed=8)                   # fragments clipped to chromosomes
    check(ja, O, tiny, 150, 3000, 10, job())
    check(ja, O, tiny, 150, 1500, 10, job(paired=False))


def test_empty_chromosomes_get_no_reads(ja, O):
    """A chromosome without bases has probability 0 in reads_per_group (src/hts.h:78 `continue`): no lane gets reads for it,
    in any position of the list.  (Refused in rounds 1-2.)"""
    rng = np.random.default_rng(21)
    seqs = [rng.choice(np.frombuffer(b"TCAG", dtype=np.uint8), size=n) for n in (0, 30_000, 0, 0, 7_000, 151, 0)]
    g = ja.RefGenome(seqs)
    check(ja, O, g, 150, 6000, 37, job())
    check(ja, O, g, 150, 3001, 5, job(paired=False, barcode="ACGTAC", prob_dup=0.3))
    with pytest.raises(ja.JackalopeHipError, match="holds no bases"):
        ja.illumina(ja.RefGenome([seqs[0], seqs[2]]), None, 10, 150, True, n_threads=1, seed_words=ja.seed_words(1, 16), _session=True)


def test_refusals_name_their_reason(ja):
    """The inputs this path refuses although the reference's C++ would take them: each ends the call
    with JK_ERR_UNSUPPORTED and a message that says what to change."""
    g = ja.synthetic_genome([20_000, 40], seed=22)
    kw = dict(n_threads=2, seed_words=ja.seed_words(1, 64), _session=True)
    # a fragment shorter than the barcode: `read_chrom_spaces[r] -= barcode.size()` wraps and `read[i] = barcode[i]` writes past
    # the string in the reference (src/hts_illumina.cpp:177-182, :391): undefined there
    with pytest.raises(ja.JackalopeHipError, match="shorter than the barcode .*chromosome of 40 bases against a barcode of 41") as e:
        ja.illumina(g, None, 100, 150, True, barcodes="ACGTA" * 8 + "C", **kw)
    assert e.value.code == 2            # JK_ERR_UNSUPPORTED
    with pytest.raises(ja.JackalopeHipError, match="shorter than the barcode .*30 bases against a barcode of 32"):
        ja.illumina(g, None, 100, 150, True, barcodes="ACGT" * 8, frag_len_min=30, **kw)
    ja.illumina(g, None, 100, 150, True, barcodes="ACGT" * 10, **kw).close()          # 40 = the short chromosome: fine


def test_many_chromosomes_use_binomial_quotas(ja, O):
    g = ja.synthetic_genome([30_000, 5_000, 80_000, 12_345, 150, 40_000, 999], seed=9)
    check(ja, O, g, 150, 20_000, 50, job())
    check(ja, O, g, 150, 20_001, 3, job(paired=False, prob_dup=0.4))


def test_non_tcag_bases(ja, O):
    rng = np.random.default_rng(10)
    seq = np.frombuffer(b"TCAGNnRYtcag-", dtype=np.uint8)[rng.integers(0, 13, size=40_000)]
    g = ja.RefGenome([seq, "C" * 25 + "N" * 150 + "T" * 25])
    check(ja, O, g, 150, 4000, 21, job())
    check(ja, O, g, 150, 2000, 4, job(matepair=True))
    # zero bytes are what the reference's FASTA reader makes of every non-TCAGN character; bytes 1..3 ride along
    low = np.frombuffer(b"TCAGTCAGN\x00\x01\x02\x03", dtype=np.uint8)[rng.integers(0, 13, size=30_000)]
    check(ja, O, ja.RefGenome([low]), 150, 3000, 9, job())
    with pytest.raises(ja.JackalopeHipError, match="0xfc-0xff"):
        ja.illumina(ja.RefGenome([np.full(1000, 0xfd, dtype=np.uint8)]), None, 10, 150, True, n_threads=1,
                    seed_words=ja.seed_words(1, 16), _session=True)


@pytest.mark.parametrize("ins,dele", [(0.05, 0.05), (0.2, 0.0), (0.0, 0.3), (0.3, 0.3), (0.0, 0.0)])
def test_indel_heavy_error_models(ja, O, ins, dele):
    g = ja.synthetic_genome([100_000], seed=11)
    check(ja, O, g, 150, 3000, 16, job(ins_prob1=ins, del_prob1=dele, ins_prob2=dele, del_prob2=ins), seed=int(ins * 100))


@pytest.mark.parametrize("read_length", [36, 100, 125, 250])
def test_other_builtin_profiles(ja, O, read_length):
    """100 bp (HiSeq 2000) and 250 bp (MiSeq) tables do not fit in LDS -> global-table kernel."""
    g = ja.synthetic_genome([70_000], seed=12)
    check(ja, O, g, read_length, 2400, 12, job(frag_mean=600.0, frag_sd=120.0))


def write_random_profile(path, n_pos, seed):
    """An ART-format profile (R/hts_illumina.R:211-262) with three to five qualities per position and base."""
    rng = np.random.default_rng(seed)
    with open(path, "w") as fh:
        for nt in "ACGT":
            for pos in range(n_pos):
                k = int(rng.integers(3, 6))
                quals = np.sort(rng.choice(np.arange(2, 42), size=k, replace=False))
                cum = np.cumsum(rng.integers(1, 1000, size=k))
                fh.write("%s\t%d\t%s\n" % (nt, pos, "\t".join(str(int(q)) for q in quals)))
                fh.write("%s\t%d\t%s\n" % (nt, pos, "\t".join(str(int(c)) for c in cum)))
    return path


@pytest.mark.parametrize("read_length", [481, 600, 992])
def test_reads_longer_than_480(ja, O, tmp_path, read_length):
    """Custom profiles may be longer than any instrument's reads (R/hts_illumina.R:211-262 takes whatever the file holds).
    Above 480 the run uses the kernels with 64-bit event-word masks (csrc/jk_illumina_kernel.h: ev_t); 992 is their limit."""
    g = ja.synthetic_genome([400_000, 9_000], seed=14)
    p1 = write_random_profile(str(tmp_path / "p1.txt"), read_length, 1)
    p2 = write_random_profile(str(tmp_path / "p2.txt"), read_length, 2)
    check(ja, O, g, read_length, 1200, 70, job(frag_mean=3.0 * read_length, frag_sd=0.5 * read_length), profiles=(p1, p2))
    # indel-heavy: events in bitmap words beyond the 16th, fragments shorter than the read
    check(ja, O, g, read_length, 800, 64, job(frag_mean=1.5 * read_length, frag_sd=0.8 * read_length, ins_prob1=0.01, del_prob1=0.02,
                                             ins_prob2=0.02, del_prob2=0.01), profiles=(p1, p2), seed=5)
    # a long barcode (more than 480 bases for the two longer reads)
    check(ja, O, g, read_length, 400, 64, job(frag_mean=3.0 * read_length, frag_sd=0.5 * read_length,
                                             barcode=("GATTACAT" * 124)[:min(read_length - 20, 700)]), profiles=(p1, p2))
    if read_length == 992:
        p3 = write_random_profile(str(tmp_path / "p3.txt"), 993, 3)
        with pytest.raises(ja.JackalopeHipError, match="above 992"):
            ja.illumina(g, None, 100, 993, False, profile1=p3, n_threads=4, seed_words=ja.seed_words(1, 64), _session=True)


def test_gamma_shapes(ja, O):
    g = ja.synthetic_genome([300_000], seed=13)
    check(ja, O, g, 150, 4000, 20, job(frag_mean=300.0, frag_sd=300.0))     # shape 1
    check(ja, O, g, 150, 4000, 20, job(frag_mean=5000.0, frag_sd=500.0))    # shape 100
    # shape < 1 (any positive frag_sd is legal, R/hts_illumina.R:658): libstdc++ then draws with shape + 1 and multiplies
    # by pow(u, 1 / shape) (random.tcc:2380-2388)
    check(ja, O, g, 150, 6000, 33, job(frag_mean=400.0, frag_sd=500.0))     # shape 0.64
    check(ja, O, g, 150, 6000, 33, job(frag_mean=300.0, frag_sd=900.0, frag_len_max=5000))     # shape 0.11


@pytest.mark.parametrize("matepair", [False, True])
def test_reference_known_answer_pairs(ja, O, tmp_path, matepair):
    """test-sequencer.R:91-161 on the GPU, against its stated expectation AND against the oracle."""
    gk = json.load(open(os.path.join(HERE, "golden", "sequencer_known_answers.json")))
    prof = write_test_profile(str(tmp_path / "test_prof.txt"))
    genome = ja.RefGenome([gk["chrom"]])
    j = job(paired=True, matepair=matepair, frag_len_min=200, frag_len_max=200, ins_prob1=0, del_prob1=0,
            ins_prob2=0, del_prob2=0)
    h1, h2 = check(ja, O, genome, 100, gk["n_reads"], 8, j, profiles=(prof, prof))
    expect = gk["matepair_expected_reads" if matepair else "paired_expected_reads"]
    for data in (h1, h2):
        assert sorted(set(r[1].decode() for r in fastq_records(data))) == expect


def test_files_written_like_the_reference(ja, O, tmp_path):
    """One-shot jk_illumina_ref: <prefix>_R1.fq / _R2.fq, overwrite policy, same bytes as a session."""
    g = ja.synthetic_genome([40_000], seed=14)
    words = ja.seed_words(5, 16 * 8)
    prefix = str(tmp_path / "test")
    ja.illumina(g, prefix, 1000, 150, True, n_threads=8, seed_words=words)
    with pytest.raises(FileExistsError):
        ja.illumina(g, prefix, 1000, 150, True, n_threads=8, seed_words=words)
    ja.illumina(g, prefix, 1000, 150, True, n_threads=8, seed_words=words, overwrite=True)
    h1, h2, _, _ = run_hip(ja, g, (None, None), 150, words, 1000, 8, job())
    assert open(prefix + "_R1.fq", "rb").read() == h1 and open(prefix + "_R2.fq", "rb").read() == h2
    lines = h1.split(b"\n")
    assert len(lines) == 2001 and all(x.startswith(b"@") for x in lines[0:2000:4]) and set(lines[2:2000:4]) == {b"+"}


def test_lane_shards_concatenate_to_the_whole(ja, O):
    """Multi-GPU sharding contract on one GPU: lane blocks generated by separate sessions (what each
    rank does) concatenate to the single-session output, and batching does not change bytes."""
    g = ja.synthetic_genome([150_000], seed=15)
    T, n = 96, 9000
    words = ja.seed_words(8, 16 * T)
    j = job()
    whole1, whole2, _, _ = run_hip(ja, g, (None, None), 150, words, n, T, j)
    parts1, parts2 = [], []
    for lo, hi in [(0, 31), (31, 32), (32, 96)]:
        a, b, _, _ = run_hip(ja, g, (None, None), 150, words, n, T, j, lane_begin=lo, lane_end=hi)
        parts1.append(a)
        parts2.append(b)
    assert b"".join(parts1) == whole1 and b"".join(parts2) == whole2
    small1, small2, _, _ = run_hip(ja, g, (None, None), 150, words, n, T, j, max_batch_bytes=200_000)
    assert small1 == whole1 and small2 == whole2
    with open_hip(ja, g, (None, None), 150, words, n, T, j) as s:
        s.generate()
        first = s.fetch(0)
        s.generate()
        assert s.fetch(0) == first == whole1          # regenerate is idempotent
        lb = s.lane_bytes(0, T)
        assert int(lb.sum()) == len(whole1)


def test_compressed_sinks(ja, O, tmp_path):
    """FileGZ / FileBGZF (src/io.h:58-236): <prefix>_R{1,2}.fq.gz whose decompressed content is the plain FASTQ."""
    import gzip
    g = ja.synthetic_genome([60_000], seed=17)
    words = ja.seed_words(4, 16 * 32)
    plain1, plain2, _, _ = run_hip(ja, g, (None, None), 150, words, 4000, 32, job())
    for method, level in (("bgzip", True), ("gzip", 1), ("bgzip", 9)):
        prefix = str(tmp_path / ("z_%s_%s" % (method, level)))
        ja.illumina(g, prefix, 4000, 150, True, n_threads=32 if method == "bgzip" else 1, seed_words=words,
                    compress=level, comp_method=method) if method == "bgzip" else None
        if method == "gzip":
            w1 = ja.seed_words(4, 16)
            ja.illumina(g, prefix, 400, 150, True, n_threads=1, seed_words=w1, compress=level, comp_method="gzip")
            p1, _, _, _ = run_hip(ja, g, (None, None), 150, w1, 400, 1, job())
            assert gzip.decompress(open(prefix + "_R1.fq.gz", "rb").read()) == p1
            continue
        raw = open(prefix + "_R1.fq.gz", "rb").read()
        assert gzip.decompress(raw) == plain1 and gzip.decompress(open(prefix + "_R2.fq.gz", "rb").read()) == plain2
        assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC" and raw[-28:-24] == b"\x1f\x8b\x08\x04"   # BGZF framing + EOF block
        assert not os.path.exists(prefix + "_R1.fq")


def test_unsupported_inputs_fail_loudly(ja):
    g = ja.synthetic_genome([10_000], seed=16)
    words = ja.seed_words(1, 64)
    # a lane's position in its pool is 32-bit: 14 M pairs on one lane would be 4.6 GiB per read end.  Refused before any
    # allocation, with the remedy in the message (R users meet this with n_threads = 1 and a large n_reads).
    with pytest.raises(ja.JackalopeHipError, match="raise n_threads") as e:
        ja.illumina(g, None, 28_000_000, 150, True, n_threads=1, seed_words=words, _session=True)
    assert e.value.code == 2
    # pools + image beyond device memory: the sizes are named instead of a bare hipMalloc failure (64 lanes x 6 M pairs:
    # every lane stays below 4 GiB, the tile's pools would need ~260 GB per read end)
    with pytest.raises(ja.JackalopeHipError, match="MiB of device memory"):
        ja.illumina(g, None, 12_000_000 * 64, 150, True, n_threads=64, seed_words=ja.seed_words(1, 16 * 64), _session=True)
    with pytest.raises(ja.JackalopeHipError, match="seed"):
        ja.illumina(g, None, 100, 150, True, n_threads=4, seed_words=words[:8], _session=True)


def test_seed_callback_and_abort_flag(ja, O):
    """The two hooks the Rcpp shim relies on: seeds pulled through a callback (Rcpp::runif there), eight words per
    request in the reference's order, and the abort flag (Progress::check_abort) polled between batches."""
    g = ja.synthetic_genome([50_000, 7_000], seed=18)
    T, n = 12, 2400
    words = ja.seed_words(21, 16 * T)
    want1, want2, _, used = run_hip(ja, g, (None, None), 150, words, n, T, job())
    calls = []

    def next8():
        k = len(calls)
        calls.append(k)
        return words[8 * k:8 * k + 8]
    with ja.illumina(g, None, n, 150, True, n_threads=T, seed_fn=next8, _session=True) as s:
        s.generate()
        assert (s.fetch(0), s.fetch(1)) == (want1, want2)
        assert s.seed_words_used() == used == 8 * len(calls)
    with pytest.raises(ja.JackalopeHipError, match="seed callback failed"):
        ja.illumina(g, None, n, 150, True, n_threads=T, seed_fn=lambda: [1, 2, 3], _session=True)
    flag = np.ones(1, dtype=np.int32)
    with ja.illumina(g, None, n, 150, True, n_threads=T, seed_words=words, abort_flag=flag, _session=True) as s:
        with pytest.raises(ja.JackalopeHipError, match="aborted"):
            s.generate()
        flag[0] = 0
        s.generate()
        assert s.fetch(0) == want1
