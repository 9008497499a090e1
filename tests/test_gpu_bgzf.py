"""Device BGZF sink (jk_bgzf_deflate; replaces FileBGZF / bgzip_file, src/io.h:150-236, src/hts.h:140-180).
Parity criterion for a compressed sink: an independent inflate (Python's zlib/gzip, which also checks
every member's CRC-32 and ISIZE) restores exactly the input bytes, and the container is well-formed BGZF
(what htslib's bgzf_read / `bgzip -d` / samtools accept): 0xff00-byte input blocks, 'BC' extra field whose
BSIZE chains the members, the fixed end-of-file block."""
import gzip
import struct
import zlib

import numpy as np
import pytest

from helpers import job

pytestmark = pytest.mark.gpu

EOF_BLOCK = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def walk_bgzf(raw):
    """Split a BGZF image into members using BSIZE only; returns [(member bytes, inflated bytes)]."""
    out, at = [], 0
    while at < len(raw):
        assert raw[at:at + 4] == b"\x1f\x8b\x08\x04", at
        xlen = struct.unpack_from("<H", raw, at + 10)[0]
        assert xlen == 6 and raw[at + 12:at + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", raw, at + 16)[0] + 1
        member = raw[at:at + bsize]
        data = zlib.decompressobj(-15).decompress(member[18:-8])
        crc, isize = struct.unpack("<II", member[-8:])
        assert isize == len(data) and crc == zlib.crc32(data)
        out.append((member, data))
        at += bsize
    assert at == len(raw)
    return out


def roundtrip(ja, data):
    comp = bytes(ja.bgzf_deflate(data).cpu().numpy().tobytes())
    assert len(comp) <= ja.bgzf_bound(len(data))
    assert gzip.decompress(comp) == bytes(data)
    members = walk_bgzf(comp)
    assert members[-1][0] == EOF_BLOCK and members[-1][1] == b""
    sizes = [len(d) for _, d in members[:-1]]
    assert all(s == 0xff00 for s in sizes[:-1]) and (not sizes or 0 < sizes[-1] <= 0xff00)
    assert sum(sizes) == len(data)
    return comp


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 1000, 0xff00 - 1, 0xff00, 0xff00 + 1, 3 * 0xff00 + 77])
def test_sizes_and_block_edges(ja, n):
    rng = np.random.default_rng(n)
    data = bytes(np.frombuffer(b"ACGTN\n@+IIIIFFF#", dtype=np.uint8)[rng.integers(0, 16, size=n)])
    roundtrip(ja, data)


def test_fastq_from_the_generator_stays_on_the_device(ja):
    g = ja.synthetic_genome([200_000], seed=71)
    words = ja.seed_words(2, 16 * 64)
    with ja.illumina(g, None, 40_000, 150, True, n_threads=64, seed_words=words, _session=True) as s:
        s.generate()
        plain = s.fetch(0)
        comp = roundtrip(ja, plain)
        ratio = len(comp) / len(plain)
        assert 0.30 < ratio < 0.45, ratio        # order-0 entropy of FASTQ, about zlib level 1


def test_incompressible_input_falls_back_to_stored_blocks(ja):
    data = np.random.default_rng(5).integers(0, 256, size=2 * 0xff00 + 123, dtype=np.uint8).tobytes()
    comp = roundtrip(ja, data)
    first = walk_bgzf(comp)[0][0]
    assert first[18] == 0x01 and len(first) == 18 + 5 + 0xff00 + 8          # BFINAL=1, BTYPE=0


def test_single_symbol_and_two_symbol_blocks(ja):
    roundtrip(ja, b"A" * 100_000)
    roundtrip(ja, b"AC" * 70_000)
    roundtrip(ja, bytes(range(256)) * 600)        # all 256 literals used, flat histogram


def test_code_length_limit(ja):
    """Fibonacci-like byte frequencies make the unrestricted Huffman tree deeper than DEFLATE's 15 bits."""
    counts, a, b = [], 1, 1
    while sum(counts) + a <= 0xff00:
        counts.append(a)
        a, b = b, a + b
    assert len(counts) >= 20
    block = b"".join(bytes([65 + i]) * c for i, c in enumerate(counts))
    rng = np.random.default_rng(3)
    shuffled = bytes(np.frombuffer(block, dtype=np.uint8)[rng.permutation(len(block))])
    roundtrip(ja, block)
    roundtrip(ja, shuffled + block)


def test_write_path_uses_it_and_host_variants_agree(ja, tmp_path):
    """jk_session_write: "bgzip" = device blocks, "bgzip-host" = zlib on the host, same decompressed FASTQ."""
    g = ja.synthetic_genome([80_000], seed=72)
    words = ja.seed_words(5, 16 * 16)
    j = job()
    with ja.illumina(g, None, 6000, 150, True, n_threads=16, seed_words=words, _session=True) as s:
        s.generate()
        plain = [s.fetch(0), s.fetch(1)]
    sizes = {}
    for method in ("bgzip", "bgzip-host"):
        prefix = str(tmp_path / method)
        ja.illumina(g, prefix, 6000, 150, True, n_threads=16, seed_words=words, compress=6, comp_method=method)
        for e in (0, 1):
            raw = open("%s_R%d.fq.gz" % (prefix, e + 1), "rb").read()
            assert gzip.decompress(raw) == plain[e]
            assert walk_bgzf(raw)[-1][0] == EOF_BLOCK
            sizes[(method, e)] = len(raw)
    assert sizes[("bgzip-host", 0)] < sizes[("bgzip", 0)] < 1.35 * sizes[("bgzip-host", 0)]
