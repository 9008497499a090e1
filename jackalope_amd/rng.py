"""Stand-in for R's RNG at the boundary.

The reference seeds every pcg64 from eight 32-bit words drawn with ``Rcpp::runif(8, 0, 2^32)``
(/root/reference/src/pcg.h:37-46,63-71).  R is not available here, so tests and benchmarks draw those
words from a SplitMix64 stream instead; the Rcpp shim in INTEGRATION.md draws them from R exactly as
the reference does.  What matters for parity is only that the oracle and the HIP path receive the
same words in the same order.
"""
import numpy as np

_M64 = (1 << 64) - 1


def seed_words(seed, n_words):
    """``n_words`` 32-bit words: the high halves of consecutive SplitMix64 outputs (made in pieces so that a
    multi-GPU run's 10^8 words do not need gigabytes of temporaries)."""
    n = int(n_words)
    out = np.empty(n, dtype=np.uint32)
    step = 1 << 24
    with np.errstate(over="ignore"):
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            idx = np.arange(lo + 1, hi + 1, dtype=np.uint64)
            z = np.uint64(seed & _M64) + idx * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            out[lo:hi] = (z >> np.uint64(32)).astype(np.uint32)
    return out


def illumina_ref_seed_budget(n_threads):
    """Upper bound on the words an illumina() run over a reference genome consumes:
    8 per lane (mt_seeds) + 8 per lane (reads_per_group inside add_n_reads)."""
    return 16 * int(n_threads)
