"""One PacBio generate() of N reads of one fixed length (argv: length, reads per lane, lanes): run under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU` to split the kernels' instruction counts into a per-read and a per-position part."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import jackalope_amd as ja
L, per, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = ja.synthetic_genome([125_000_000] * 4, seed=3)
words = ja.seed_words(5, 16 * T)
s = ja.pacbio(g, None, T * per, n_threads=T, seed_words=words, _session=True, custom_read_lengths=[L])
with s:
    s.generate()
    sizes, reads = s.sizes()
    print("reads", reads, "bytes", sizes[0], "launches", s.n_batches(), flush=True)
