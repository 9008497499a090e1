"""Quick PacBio throughput probe (not the headline bench): mean-10kb reads on a synthetic genome."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jackalope_amd as ja

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
g = ja.synthetic_genome([int(mbp * 1e6)], seed=3)
n_reads = int(mbp * 1e6 * 20 / 10000)
lens = list(range(5000, 15001, 500))
words = ja.seed_words(12345, 16 * lanes)
t = time.time()
s = ja.pacbio(g, None, n_reads, n_threads=lanes, seed_words=words, custom_read_lengths=lens, _session=True)
print("open %.2fs" % (time.time() - t))
with s:
    for i in range(3):
        t = time.time(); s.generate(); dt = time.time() - t
        sizes, reads = s.sizes(); tm = s.timing_ms()
        print("reads %d bytes %.2f GB  wall %.1f ms  gen %.1f ms  -> %.3f M reads/s, %.2f Gbases/s, batches %d" % (
            reads, sizes[0] / 1e9, dt * 1e3, tm["generate_kernel"], reads / dt / 1e6, sizes[0] / 2 / dt / 1e9, s.n_batches()))
