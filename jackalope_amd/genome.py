"""Inputs of the read-generation path: reference genomes (and, later, haplotype sets).

These mirror only what the sequencers *read* from jackalope's objects: ``RefGenome``
(/root/reference/src/ref_classes.h:127-180: name "REF", chromosomes {name, nucleos}).  Genome
creation/evolution are out of scope; ``synthetic_genome`` is this repo's own seeded generator for
tests and benchmarks.
"""
import ctypes as C

import numpy as np

from . import _abi


class RefGenome:
    """Named chromosomes held as ASCII bytes (one byte per base, as the reference stores them)."""

    def __init__(self, seqs, names=None, name="REF"):
        self.seqs = [np.frombuffer(s.encode() if isinstance(s, str) else bytes(s), dtype=np.uint8)
                     if not isinstance(s, np.ndarray) else np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
        # make_ref_genome names chromosomes chrom0.. (/root/reference/src/ref_hap_access.cpp:109-113)
        self.names = list(names) if names is not None else ["chrom%d" % i for i in range(len(self.seqs))]
        if len(self.names) != len(self.seqs):
            raise ValueError("names and seqs differ in length")
        self.name = name

    def n_chroms(self):
        return len(self.seqs)

    def sizes(self):
        return [int(s.size) for s in self.seqs]

    def _view(self):
        """(jk_ref_genome struct, keep-alive list)"""
        n = len(self.seqs)
        names = (C.c_char_p * n)(*[x.encode() for x in self.names])
        ptrs = (C.c_void_p * n)(*[s.ctypes.data for s in self.seqs])
        lens = (C.c_uint64 * n)(*[s.size for s in self.seqs])
        v = _abi.RefGenomeView(n, names, ptrs, lens, self.name.encode())
        return v, [names, ptrs, lens, self.seqs]


def synthetic_genome(chrom_sizes, seed, alphabet=b"TCAG"):
    """iid-uniform chromosomes over ``alphabet`` from numpy's seeded generator (not the reference's
    create_genome; the sequencers only need *a* genome)."""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(alphabet, dtype=np.uint8)
    seqs = [lut[rng.integers(0, lut.size, size=int(n), dtype=np.uint8)] for n in chrom_sizes]
    return RefGenome(seqs)
