"""PacBio generate() time on three shapes of job (reads/s): long custom reads as the bench, the default log-normal lengths,
many short reads per lane.   usage: pb_workloads_probe.py   (library chosen by JK_HIP_LIB, slots by JK_PB_WAVES_PER_CU)"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import jackalope_amd as ja
g = ja.synthetic_genome([125_000_000] * 8, seed=3)
jobs = [("uniform 5-15 kb, 2^21 lanes x 3", 1 << 21, 3, {"custom_read_lengths": list(range(5000, 15001, 100))}),
        ("default log-normal, 2^20 lanes x 6", 1 << 20, 6, {}),
        ("short 300-1500, 2^19 lanes x 40", 1 << 19, 40, {"custom_read_lengths": list(range(300, 1501, 50))}),
        ("uniform 5-15 kb, 2^14 lanes x 100", 1 << 14, 100, {"custom_read_lengths": list(range(5000, 15001, 100))})]
for name, T, per, pb in jobs:
    words = ja.seed_words(5, 16 * T)
    s = ja.pacbio(g, None, T * per, n_threads=T, seed_words=words, _session=True, **pb)
    with s:
        s.generate()
        ts = []
        for _ in range(3):
            t0 = time.time(); s.generate(); ts.append(time.time() - t0)
        sizes, reads = s.sizes()
        print("%-40s %8.2f M reads/s  %7.1f ms  %5.1f GB  %d launches" % (name, reads / min(ts) / 1e6, min(ts) * 1e3, sizes[0] / 1e9, s.n_batches()), flush=True)
