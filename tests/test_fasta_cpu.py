"""The oracle's restatement of the reference FASTA reader (src/io_fasta.cpp:41-408) on hand-made files whose
outcome follows from the reference's rules, and on the reference's own test shape (tests/testthat/test-fasta_IO.R:
write 10 chromosomes of 100 bp at text_width 80, read back, identical)."""
import gzip

import numpy as np

from helpers import write_fasta, write_fai


def test_handmade_rules(O, tmp_path):
    fn = str(tmp_path / "a.fa")
    open(fn, "wb").write(b">chr1 first one\r\nACGTacgt\r\nNNnnRY-*\n\n>two\nTTTT\n>empty\n>last x\nGG\rA\nC")
    names, seqs = O.read_fasta([fn])
    assert names == [b"chr1 first one", b"two", b"empty", b"last x"]
    # CR before LF dropped, soft mask removed, other bytes (R, Y, -, *, lone CR) -> zero bytes, blank line adds nothing
    assert seqs == [b"ACGTACGTNNNN\0\0\0\0", b"TTTT", b"", b"GG\0AC"]
    names, _ = O.read_fasta([fn], cut_names=True)
    assert names == [b"chr1", b"two", b"empty", b"last"]
    _, seqs = O.read_fasta([fn], remove_soft_mask=False)
    assert seqs[0] == b"ACGTacgtNNnn\0\0\0\0"


def test_reference_roundtrip_shape(O, tmp_path):
    rng = np.random.default_rng(1)
    chroms = [bytes(rng.choice(np.frombuffer(b"TCAG", dtype=np.uint8), size=100)) for _ in range(10)]
    names = ["chrom%d" % i for i in range(10)]
    fn = str(tmp_path / "t.fa")
    write_fasta(fn, names, chroms)
    got_names, got = O.read_fasta([fn])
    assert got == chroms and got_names == [n.encode() for n in names]
    gz = fn + ".gz"
    open(gz, "wb").write(gzip.compress(open(fn, "rb").read()))
    assert O.read_fasta([gz])[1] == chroms
    fai = write_fai(fn + ".fai", names, chroms)
    got_names, got = O.read_fasta([fn], [fai])
    assert got == chroms and got_names == [n.encode() for n in names]
    assert O.read_fasta([fn, gz])[1] == chroms + chroms            # several files append
