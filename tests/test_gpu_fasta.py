"""GPU parity for read_fasta (read_fasta_noind / read_fasta_ind, src/io_fasta.cpp:153-169, :389-408): the device
packs the raw text (newline removal, filter table, per-chromosome offsets); the oracle restates the reference's
buffered line parser.  Names and chromosome bytes must be identical; cases follow tests/testthat/test-fasta_IO.R
(single/multiple files; plain, gzip, bgzip; indexed and not) plus the parser's edge rules."""
import gzip

import numpy as np
import pytest

from helpers import write_fasta, write_fai, job

pytestmark = pytest.mark.gpu


def same(ja, O, files, fais=None, cut_names=False):
    want_names, want = O.read_fasta(files, fais, cut_names=cut_names)
    g = ja.read_fasta(files, fais, cut_names=cut_names)
    assert [n.encode() for n in g.names] == want_names
    assert g.sizes() == [len(c) for c in want]
    for i, w in enumerate(want):
        assert g.chrom(i).tobytes() == w, i
    return g


def random_chroms(seed, sizes, alphabet=b"TCAG"):
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(alphabet, dtype=np.uint8)
    return [bytes(lut[rng.integers(0, lut.size, size=n)]) for n in sizes]


def test_reference_shape_all_containers(ja, O, tmp_path):
    chroms = random_chroms(1, [100] * 10)
    names = ["chrom%d" % i for i in range(10)]
    fn = str(tmp_path / "t.fa")
    write_fasta(fn, names, chroms)
    g = same(ja, O, [fn])
    assert [g.chrom(i).tobytes() for i in range(10)] == chroms
    gz = str(tmp_path / "t_gz.fa.gz")
    open(gz, "wb").write(gzip.compress(open(fn, "rb").read()))
    same(ja, O, [gz])
    bgz = str(tmp_path / "t_bgz.fa.gz")
    open(bgz, "wb").write(ja.bgzf_deflate(open(fn, "rb").read()).cpu().numpy().tobytes())     # this library's own BGZF
    same(ja, O, [bgz])
    fai = write_fai(fn + ".fai", names, chroms)
    same(ja, O, [fn], [fai])
    same(ja, O, [gz], [fai])
    # multiple files (test-fasta_IO.R:141-210)
    f1, f2 = str(tmp_path / "p1.fa"), str(tmp_path / "p2.fa")
    write_fasta(f1, names[:5], chroms[:5]); write_fasta(f2, names[5:], chroms[5:])
    g = same(ja, O, [f1, f2])
    assert [g.chrom(i).tobytes() for i in range(10)] == chroms
    same(ja, O, [f1, f2], [write_fai(f1 + ".fai", names[:5], chroms[:5]), write_fai(f2 + ".fai", names[5:], chroms[5:])])


def test_parser_edge_rules(ja, O, tmp_path):
    fn = str(tmp_path / "edge.fa")
    open(fn, "wb").write(b">chr1 first one\r\nACGTacgt\r\nNNnnRY-*\n\n>two\nTTTT\n>empty\n>last x\nGG\rA\nC")
    g = same(ja, O, [fn])
    assert g.names == ["chr1 first one", "two", "empty", "last x"]
    assert g.chrom(0).tobytes() == b"ACGTACGTNNNN\0\0\0\0" and g.sizes()[2] == 0
    same(ja, O, [fn], cut_names=True)
    f2 = str(tmp_path / "edge2.fa")
    open(f2, "wb").write(b">a\nAC>GT\nTT\n>only header at the end")        # a '>' inside a line starts a chromosome too
    g = same(ja, O, [f2])
    assert g.names == ["a", "C>GT", "only header at the end"] and g.sizes() == [0, 2, 0]
    f3 = str(tmp_path / "crlf.fa")
    chroms = random_chroms(3, [333, 80, 81, 1])
    write_fasta(f3, ["w%d" % i for i in range(4)], chroms, text_width=60, newline=b"\r\n")
    g = same(ja, O, [f3])
    assert [g.chrom(i).tobytes() for i in range(4)] == chroms


@pytest.mark.parametrize("sizes,width", [([1_000_000, 4096, 4095, 4097, 15, 16, 17], 80), ([300_000] * 3, 1), ([2_500_000], 100_000)])
def test_block_boundaries_and_line_widths(ja, O, tmp_path, sizes, width):
    chroms = random_chroms(len(sizes) + width, sizes, alphabet=b"TCAGNtcagn")
    names = ["s%d some description" % i for i in range(len(sizes))]
    fn = str(tmp_path / "big.fa")
    write_fasta(fn, names, chroms, text_width=width)
    g = same(ja, O, [fn])
    assert [g.chrom(i).tobytes() for i in range(len(sizes))] == [c.upper() for c in chroms]
    if width > 1:
        same(ja, O, [fn], [write_fai(fn + ".fai", names, chroms, text_width=width)])


def test_unordered_index_and_sequencing_from_the_file(ja, O, tmp_path):
    chroms = random_chroms(9, [30_000, 12_000, 50_000])
    names = ["x", "y", "z"]
    fn = str(tmp_path / "g.fa")
    write_fasta(fn, names, chroms)
    fai = write_fai(fn + ".fai", names, chroms)
    lines = open(fai).read().splitlines()
    open(fai, "w").write("\n".join([lines[2], lines[0], lines[1]]) + "\n")        # index lists z, x, y
    g = same(ja, O, [fn], [fai])
    assert g.names == ["z", "x", "y"]
    # the packed genome feeds illumina() in place
    host = ja.RefGenome([g.chrom(i) for i in range(3)], names=g.names)
    T, n = 8, 2000
    words = ja.seed_words(4, 16 * T)
    with ja.illumina(g, None, n, 150, True, n_threads=T, seed_words=words, _session=True) as s:
        s.generate()
        dev = (s.fetch(0), s.fetch(1))
    with ja.illumina(host, None, n, 150, True, n_threads=T, seed_words=words, _session=True) as s:
        s.generate()
        assert dev == (s.fetch(0), s.fetch(1))


def test_iupac_codes_read_and_sequence_like_the_reference(ja, O, tmp_path):
    """Non-TCAGN characters become zero bytes in read_fasta (src/str_manip.h:24-56); the sequencers then see a
    non-TCAG base ('N' with a random quality in Illumina reads, copied through by PacBio)."""
    rng = np.random.default_rng(12)
    seq = bytes(np.frombuffer(b"TCAGTCAGTCAGNRYMKtcag", dtype=np.uint8)[rng.integers(0, 21, size=60_000)])
    fn = str(tmp_path / "iupac.fa")
    write_fasta(fn, ["c1"], [seq])
    g = same(ja, O, [fn])
    host = ja.RefGenome([g.chrom(0)], names=g.names)
    assert 0 in host.seqs[0]
    T, n = 8, 1500
    words = ja.seed_words(8, 16 * T)
    p1, p2 = ja.read_profile(None, None, 150, 1), ja.read_profile(None, None, 150, 2)
    j = job()
    o1, o2, _ = O.illumina_ref(host, paired=True, n_reads=n, prob_dup=j["prob_dup"], n_threads=T, read_pool_size=1000,
                               shape=16.0, scale=25.0, fmin=150, fmax=2 ** 32 - 1, prof1=p1, prof2=p2, ins1=j["ins_prob1"],
                               del1=j["del_prob1"], ins2=j["ins_prob2"], del2=j["del_prob2"], words=words)
    with ja.illumina(g, None, n, 150, True, n_threads=T, seed_words=words, _session=True) as s:
        s.generate()
        assert (s.fetch(0), s.fetch(1)) == (o1, o2)


def test_errors(ja, tmp_path):
    with pytest.raises(ValueError, match="argument `fasta_files` must be a character vector"):
        ja.read_fasta([])
    with pytest.raises(ValueError, match="argument `fai_files` must be NULL or a character vector of the same length"):
        ja.read_fasta(["a.fa", "b.fa"], ["a.fai"])
    with pytest.raises(ValueError, match="argument `cut_names` must be a single logical"):
        ja.read_fasta("a.fa", cut_names="yeah")
    with pytest.raises(ja.JackalopeHipError, match="gzopen of .* failed"):
        ja.read_fasta(str(tmp_path / "missing.fa"))
    bad = str(tmp_path / "bad.fa")
    open(bad, "wb").write(b"ACGT\n>late\nAC\n")
    with pytest.raises(ja.JackalopeHipError, match="before the first"):
        ja.read_fasta(bad)
