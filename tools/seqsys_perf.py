"""Illumina throughput probe for other sequencing systems / read lengths than the headline one (not the bench):
which error profiles fit in LDS and what the L2-table variant of the generator costs.

    python tools/seqsys_perf.py [seq_sys read_length [paired(0/1)]] ...      e.g.  HS25 150 1  MSv3 250 1  HS20 100 1
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jackalope_amd as ja  # noqa: E402

args = sys.argv[1:] or ["HS25", "150", "1", "MSv3", "250", "1", "HS20", "100", "1", "GA2", "75", "1", "HS25", "125", "1", "HS25", "125", "0"]
lanes = 1 << 20
genome = ja.synthetic_genome([100_000_000], seed=2)
words = ja.seed_words(12345, 16 * lanes)
for k in range(0, len(args), 3):
    sys_name, L, paired = args[k], int(args[k + 1]), bool(int(args[k + 2]))
    n_pairs = int(100e6 * 30 / (L * (2 if paired else 1)))
    n_reads = n_pairs * (2 if paired else 1)
    s = ja.illumina(genome, None, n_reads, L, paired, seq_sys=sys_name, n_threads=lanes, seed_words=words, _session=True)
    with s:
        best = None
        for _ in range(3):
            t = time.time(); s.generate(); dt = time.time() - t
            best = dt if best is None or dt < best else best
        sizes, reads = s.sizes(); tm = s.timing_ms()
        print("%-5s L=%3d %s: %9d reads, %.2f GB FASTQ, %.1f ms (generator kernels %.1f ms, %d launches) -> %.1f M reads/s, %.1f Gbases/s"
              % (sys_name, L, "PE" if paired else "SE", reads, sum(sizes) / 1e9, best * 1e3, tm["generate_kernel"], s.n_batches(),
                 reads / best / 1e6, reads * L / best / 1e9), flush=True)
