"""The one-shot PacBio call fanned out over two "devices" that are both device 0 (two host threads, two sessions, one GPU),
many times over, against the oracle: tests/test_gpu_stream.py::test_pacbio_streams_and_fans_out failed ONCE in round 3 with a
valid gzip file of different content and has not failed since.  On a mismatch: where, and what the bytes look like.
usage: fanout_stress.py [iterations] [compress]"""
import gzip
import os
import sys
import tempfile
import time

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np

import jackalope_amd as ja
import oracle_lib as O

O.lib()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
compress = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ref = ja.synthetic_genome([600_000], seed=24)
n_reads, T = 3000, 512
words = ja.seed_words(80, 16 * T)
o, _, _ = O.pacbio_ref(ref, {}, n_reads=n_reads, n_threads=T, words=words)
oa = np.frombuffer(o, dtype=np.uint8)
bad = 0
t0 = time.time()
with tempfile.TemporaryDirectory() as d:
    for it in range(iters):
        b = os.path.join(d, "b%d" % it)
        ja.pacbio(ref, b, n_reads, n_threads=T, seed_words=words, devices=[0, 0], compress=compress if compress else False)
        fn = b + "_R1.fq" + (".gz" if compress else "")
        raw = open(fn, "rb").read()
        got = gzip.decompress(raw) if compress else raw
        os.unlink(fn)
        if got != o:
            bad += 1
            ga = np.frombuffer(got, dtype=np.uint8)
            m = min(ga.size, oa.size)
            dd = np.nonzero(ga[:m] != oa[:m])[0]
            print("iteration %d: lengths %d / %d, %d differing bytes in the common part, first %d last %d" % (it, ga.size, oa.size, dd.size, dd[0] if dd.size else -1, dd[-1] if dd.size else -1), flush=True)
            if dd.size:
                i = int(dd[0])
                print("   got    %r\n   oracle %r" % (got[max(i - 60, 0):i + 100], o[max(i - 60, 0):i + 100]), flush=True)
                # runs of differences
                brk = np.nonzero(np.diff(dd) > 64)[0]
                print("   %d separate regions; first region %d..%d" % (brk.size + 1, dd[0], dd[brk[0]] if brk.size else dd[-1]), flush=True)
        if it % 25 == 24:
            print("%d iterations, %d bad, %.1f s" % (it + 1, bad, time.time() - t0), flush=True)
print("RESULT compress=%d: %d of %d differ" % (compress, bad, iters))
