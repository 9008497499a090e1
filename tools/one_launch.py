"""The headline Illumina shape as a handful of full-chip launches and nothing else (a target for rocprofv3 runs).

    python tools/one_launch.py [repeats]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jackalope_amd as ja  # noqa: E402

lanes = int(os.environ.get("LANES", 1 << 18))
genome = ja.synthetic_genome([100_000_000], seed=2)
words = ja.seed_words(12345, 16 * lanes)
s = ja.illumina(genome, None, 5_000_000, 150, True, seq_sys="HS25", n_threads=lanes, seed_words=words, _session=True)
with s:
    best = None
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        s.generate()
        k = s.timing_ms()["generate_kernel"]
        best = k if best is None or k < best else best
    print("generator %.3f ms (best of the repeats; %s)" % (best, os.environ.get("JK_HIP_LIB", "product library")))
