"""Mutation-table builder (jk_add_substitution / jk_add_insertion / jk_add_deletion, restating
HapChrom::add_*, /root/reference/src/hap_classes.cpp:295-509), CPU only.

Pinned two ways, both taken from the reference's own tests:
 * tests/testthat/test-vcf_IO.R:18-90 -- a fixed list of edits on "TCAGTCAGTC" whose outcome the
   reference states as a VCF table (REF/ALT alleles and per-haplotype genotypes);
 * tests/testthat/test-R_classes.R:196-243 -- random edits compared with plain string editing.
The resulting tables must also read back identically through the three independent readers
(library get_chrom_full, oracle get_chrom_full, HapSet.materialize)."""
import ctypes as C

import numpy as np
import pytest

from jackalope_amd import _abi
from jackalope_amd.genome import HapBuilder, RefGenome, random_haplotypes, synthetic_genome


def apply_vcf(ref, rows, hap):
    """Haplotype string from (pos, REF, ALT list, genotypes) rows: allele 0 keeps REF."""
    out, at = [], 0
    for pos, ref_al, alts, gts in rows:
        out.append(ref[at:pos - 1])
        assert ref[pos - 1:pos - 1 + len(ref_al)] == ref_al
        g = gts[hap]
        out.append(ref_al if g == 0 else alts[g - 1])
        at = pos - 1 + len(ref_al)
    out.append(ref[at:])
    return "".join(out)


def test_vcf_io_fixture(built):
    """Edits of test-vcf_IO.R:18-66, expectation from its vcf_info table (:71-88)."""
    ref = RefGenome(["TCAGTCAGTC"] * 2)
    b = HapBuilder(ref, 4)
    c = 1
    b.add_sub(1, c, 6, "T"); b.add_sub(2, c, 6, "A"); b.add_sub(3, c, 6, "A"); b.add_sub(3, c, 7, "T")
    b.add_sub(4, c, 6, "G"); b.add_sub(4, c, 8, "T")
    b.add_del(1, c, 7, 1); b.add_del(2, c, 7, 2); b.add_del(4, c, 9, 1)
    b.add_del(1, c, 1, 3); b.add_del(2, c, 2, 3); b.add_del(3, c, 1, 3); b.add_del(4, c, 3, 2)
    c = 2
    b.add_del(1, c, 9, 1)
    b.add_ins(1, c, 8, "A")
    b.add_sub(1, c, 6, "A"); b.add_sub(2, c, 6, "A"); b.add_sub(3, c, 6, "T"); b.add_del(4, c, 6, 1)
    b.add_ins(1, c, 5, "TT"); b.add_ins(2, c, 5, "TT"); b.add_ins(3, c, 5, "T"); b.add_ins(4, c, 5, "C")
    b.add_sub(4, c, 3, "T")
    b.add_ins(2, c, 2, "AG"); b.add_del(3, c, 2, 2); b.add_ins(4, c, 2, "AG")
    b.add_del(1, c, 1, 1)

    vcf = {
        1: [(1, "TCAG", ["G", "T", "TC"], [1, 2, 1, 3]),
            (6, "CAGT", ["TGT", "AT", "ATGT", "GAT"], [1, 2, 3, 4])],
        2: [(1, "TCA", ["CA", "TCAGA", "T", "TCAGT"], [1, 2, 3, 4]),
            (5, "TC", ["TTTA", "TTT"], [1, 1, 2, 0]),
            (8, "GT", ["GA"], [1, 0, 0, 0])],
    }
    for chrom in (1, 2):
        for hap in range(4):
            want = apply_vcf("TCAGTCAGTC", vcf[chrom], hap)
            assert b.chrom(hap + 1, chrom) == want, (chrom, hap)
            assert b.sizes(hap + 1)[chrom - 1] == len(want)


def random_edits(b, strings, rng, n_muts, max_indel=10):
    """The loop of test-R_classes.R:199-238: the same edit applied to the builder and to a Python string."""
    for h in range(b.n_haps()):
        for c in range(b.n_chroms()):
            s = strings[h][c]
            m = 0
            while m < n_muts and len(s) > 0:
                pos = int(rng.random() * len(s)) + 1
                r = rng.random()
                if r < 0.5:
                    nt = "TCAG"[int(rng.integers(0, 4))]
                    b.add_sub(h + 1, c + 1, pos, nt)
                    s = s[:pos - 1] + nt + s[pos:]
                elif r < 0.75:
                    size = min(int(rng.exponential(0.5) + 1.0), max_indel)
                    nts = "".join("TCAG"[int(i)] for i in rng.integers(0, 4, size=size))
                    b.add_ins(h + 1, c + 1, pos, nts)
                    s = s[:pos] + nts + s[pos:]
                else:
                    size = min(int(rng.exponential(0.5) + 1.0), max_indel)
                    b.add_del(h + 1, c + 1, pos, size)     # clipped at the chromosome end by the builder
                    s = s[:pos - 1] + s[pos - 1 + size:]
                m += 1
            strings[h][c] = s


@pytest.mark.parametrize("seed,sizes,n_muts,max_indel", [
    (0, [100] * 10, 100, 10),        # the reference's own shape: 10 chromosomes of 100 bp, 100 edits each
    (1, [30, 7, 1, 2], 200, 10),     # tiny chromosomes: edits pile up, chromosomes may vanish
    (2, [400], 3000, 40),            # long indels that swallow whole earlier records
    (3, [5000, 1200], 1500, 10),
])
def test_random_edits_match_string_editing(O, built, seed, sizes, n_muts, max_indel):
    ref = synthetic_genome(sizes, seed=seed)
    n_haps = 3
    b = HapBuilder(ref, n_haps)
    strings = [[s.tobytes().decode() for s in ref.seqs] for _ in range(n_haps)]
    random_edits(b, strings, np.random.default_rng(1000 + seed), n_muts, max_indel)
    hs = b.snapshot()
    for h in range(n_haps):
        assert b.sizes(h + 1) == [len(s) for s in strings[h]]
        for c in range(len(sizes)):
            want = strings[h][c]
            assert b.chrom(h + 1, c + 1) == want, (h, c)
            assert hs.materialize(h, c).decode() == want
            assert O.hap_chrom_full(hs, h, c).decode() == want
            cell = hs.cells[h][c]
            assert cell["new_pos"] == sorted(cell["new_pos"])
            assert all(0 <= p < len(ref.seqs[c]) for p in cell["old_pos"])


def test_resume_from_existing_tables(built):
    """jk_hap_builder_from: tables made elsewhere keep accepting edits."""
    ref = synthetic_genome([600, 90], seed=8)
    hs = random_haplotypes(ref, 2, seed=9, sub_rate=0.03, ins_rate=0.02, del_rate=0.02)
    b = HapBuilder.from_hapset(hs)
    strings = [[hs.materialize(h, c).decode() for c in range(2)] for h in range(2)]
    assert [[b.chrom(h + 1, c + 1) for c in range(2)] for h in range(2)] == strings
    random_edits(b, strings, np.random.default_rng(5), 300)
    assert [[b.chrom(h + 1, c + 1) for c in range(2)] for h in range(2)] == strings
    assert b.snapshot().hap_names() == hs.hap_names()


def test_substitution_back_to_reference_removes_the_record(built):
    ref = RefGenome(["TCAGTCAGTC"])
    b = HapBuilder(ref, 1)
    b.add_sub(1, 1, 4, "A")
    assert b.snapshot().cells[0][0]["nucleos"] == ["A"]
    b.add_sub(1, 1, 4, "G")          # the reference base: src/hap_classes.cpp:493-496
    assert b.snapshot().cells[0][0]["new_pos"] == []
    assert b.chrom(1, 1) == "TCAGTCAGTC"


def test_deletion_is_clipped_and_can_empty_a_chromosome(built):
    ref = RefGenome(["TCAGTCAGTC"])
    b = HapBuilder(ref, 1)
    b.add_del(1, 1, 8, 100)
    assert b.chrom(1, 1) == "TCAGTCA"
    b.add_del(1, 1, 1, 7)
    assert b.chrom(1, 1) == "" and b.sizes(1) == [0]
    with pytest.raises(ValueError, match="argument `pos` must be integer in range"):
        b.add_sub(1, 1, 1, "A")      # check_pos: nothing left to edit


def test_argument_checks(built):
    """haplotypes$add_sub/add_ins/add_del argument errors (R/aaa-classes.R:791-850) and the C ABI's own."""
    ref = RefGenome(["TCAGTCAGTC", "AAAA"])
    b = HapBuilder(ref, 2)
    with pytest.raises(ValueError, match="`add_sub` function in jackalope, argument `pos`"):
        b.add_sub(1, 1, 11, "A")
    with pytest.raises(ValueError, match="argument `pos`"):
        b.add_ins(1, 2, 0, "A")
    with pytest.raises(ValueError, match="argument `chrom_ind`"):
        b.add_del(1, 3, 1, 1)
    with pytest.raises(ValueError, match="argument `hap_ind`"):
        b.add_sub(3, 1, 1, "A")
    with pytest.raises(ValueError, match="argument `nt` must be a single character"):
        b.add_sub(1, 1, 1, "AC")
    with pytest.raises(ValueError, match='argument `nt` must be one of "T", "C", "A", "G", or "N"'):
        b.add_sub(1, 1, 1, "X")
    with pytest.raises(ValueError, match="argument `nts`"):
        b.add_ins(1, 1, 1, "AXC")
    with pytest.raises(ValueError, match="argument `n_nts` must be a single integer >= 1"):
        b.add_del(1, 1, 1, 0)
    L = _abi.lib()
    assert L.jk_add_substitution(b._h, 0, 0, b"A", 10) == _abi.JK_ERR_ARG
    assert b"new_pos should never be >= the chromosome size" in L.jk_last_error()
    assert L.jk_add_insertion(b._h, 0, 5, b"A", 0) == _abi.JK_ERR_ARG
    assert L.jk_add_deletion(b._h, 0, 0, 0, 3) == _abi.JK_OK       # size 0: silent no-op
    assert L.jk_add_deletion(b._h, 0, 0, 2, 10) == _abi.JK_OK      # past the end: silent no-op
    assert b.chrom(1, 1) == "TCAGTCAGTC"
    h = C.c_void_p()
    assert L.jk_hap_builder_new(None, 1, C.byref(h)) == _abi.JK_ERR_ARG
