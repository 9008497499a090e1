"""Illumina throughput on a genome with runs of N (not the bench): 100 Mbp, PE150 at 30x, `frac` of the bases in
10 blocks of N, as in a real assembly with gaps.  Reads that fall into a block are all 'N' (1 draw per base instead of
3), and with 64 lanes per wave most waves hold such a lane at any time.

    python tools/n_perf.py [frac=0.05] [illumina|pacbio]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import jackalope_amd as ja  # noqa: E402

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
n, lanes, pairs = 100_000_000, 1 << 20, 10_000_000
g = ja.synthetic_genome([n], seed=2)
seq = g.seqs[0]
blk = int(n * frac / 10)
for b in range(10):
    at = int((b + 0.5) * n / 10)
    seq[at:at + blk] = ord("N")
words = ja.seed_words(12345, 16 * lanes)
if len(sys.argv) > 2 and sys.argv[2] == "pacbio":
    n_reads = int(n * 20 / 10000)
    s = ja.pacbio(g, None, n_reads, n_threads=1 << 17, seed_words=ja.seed_words(12345, 16 << 17), custom_read_lengths=list(range(5000, 15001, 500)), _session=True)
    with s:
        for _ in range(3):
            t = time.time(); s.generate(); dt = time.time() - t
        sizes, reads = s.sizes()
        print("N fraction %.3f: %d PacBio reads in %.1f ms -> %.2f M reads/s" % (frac, reads, dt * 1e3, reads / dt / 1e6))
    sys.exit(0)
s = ja.illumina(g, None, 2 * pairs, 150, True, n_threads=lanes, seed_words=words, _session=True)
with s:
    for _ in range(4):
        t = time.time(); s.generate(); dt = time.time() - t
    sizes, reads = s.sizes(); tm = s.timing_ms()
    print("N fraction %.3f: %d pairs in %.1f ms (generator kernels %.1f ms) -> %.1f M pairs/s" % (frac, reads // 2, dt * 1e3, tm["generate_kernel"], reads / 2 / dt / 1e6))
