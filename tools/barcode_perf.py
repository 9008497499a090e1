import sys, time, os
sys.path.insert(0, os.getcwd())
import jackalope_amd as ja
lanes, pairs = 1 << 20, 10_000_000
g = ja.synthetic_genome([100_000_000], seed=2)
words = ja.seed_words(12345, 16 * lanes)
for bc in [None, "ACGTACGT", "ACGTAC"]:
    s = ja.illumina(g, None, 2 * pairs, 150, True, n_threads=lanes, seed_words=words, barcodes=bc, _session=True)
    with s:
        for _ in range(4):
            t = time.time(); s.generate(); dt = time.time() - t
        print("barcode %s: %.1f ms -> %.1f M pairs/s" % (bc, dt * 1e3, pairs / dt / 1e6))
