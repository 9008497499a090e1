"""Where does the Illumina generator's time go?  One full-chip launch (2^18 lanes) of the headline shape, re-run with
parts of the work switched off or scaled through the public parameters: per-read versus per-base cost (read length
sweep at a fixed number of reads), indel events, duplicates, single end.

    python tools/ablate_illumina.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jackalope_amd as ja  # noqa: E402

lanes = 1 << 18
genome = ja.synthetic_genome([100_000_000], seed=2)
words = ja.seed_words(12345, 16 * lanes)
n_pairs = 2_500_000


def run(label, L=150, paired=True, **kw):
    n_reads = n_pairs * (2 if paired else 1)
    s = ja.illumina(genome, None, n_reads, L, paired, seq_sys="HS25", n_threads=lanes, seed_words=words, _session=True, **kw)
    with s:
        best = None
        for _ in range(4):
            s.generate()
            k = s.timing_ms()["generate_kernel"]
            best = k if best is None or k < best else best
        sizes, reads = s.sizes()
    print("%-34s L=%3d %s: generator %.3f ms  -> %.2f ns per lane-read-end, %.3f ns per lane-base" %
          (label, L, "PE" if paired else "SE", best, best * 1e6 / (reads / lanes), best * 1e6 / (reads * L / lanes)), flush=True)
    return best


base = run("headline")
for L in (50, 100, 125):
    run("shorter reads", L=L)
run("no indels", ins_prob1=0, del_prob1=0, ins_prob2=0, del_prob2=0)
run("no duplicates", prob_dup=0)
run("single end", paired=False)
run("single end, no indels", paired=False, ins_prob1=0, del_prob1=0)
run("L=148 (whole quads)", L=148)
run("L=152 (whole 8-blocks)", L=144)
