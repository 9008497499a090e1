"""PacBio configs[4] on one GPU with a chosen launch size and pool cap: does a launch of two workgroups per slot (the
second dealt dynamically as the first retire) beat two launches?  Do more reads per lane (fewer lanes) shorten the
under-occupied end of a launch?    python tools/pacbio_launch_probe.py <batch lanes> <cap GB> [lanes = 2^21]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
batch, cap = int(sys.argv[1]), float(sys.argv[2])
os.environ["JK_BATCH_LANES"] = str(batch)
import jackalope_amd as ja  # noqa: E402

lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 21
genome = ja.synthetic_genome([3_000_000_000], seed=3)
lens = list(range(5000, 15001, 500))
words = ja.seed_words(12345, 16 * lanes)
s = ja.pacbio(genome, None, 6_000_000, n_threads=lanes, seed_words=words, custom_read_lengths=lens, max_batch_bytes=int(cap * 2 ** 30), _session=True)
with s:
    best = None
    for _ in range(3):
        t = time.time(); s.generate(); dt = time.time() - t
        best = dt if best is None or dt < best else best
    sizes, reads = s.sizes()
    print("lanes %d, batch lanes %d, cap %.0f GB: %d launches, %.1f ms -> %.2f M reads/s" % (lanes, batch, cap, s.n_batches(), best * 1e3, reads / best / 1e6), flush=True)
