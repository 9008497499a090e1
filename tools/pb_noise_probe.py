"""Does a kernel launch by SOMEBODY ELSE on the same device disturb the PacBio plan kernel's scalar stores?  A dispatch
starts with a cache invalidate; if that drops dirty lines of the scalar data cache (the plan kernel's event masks sit
there between s_store_dwordx4 and s_dcache_wb), a concurrent session -- or any other user of the GPU -- corrupts reads.
Runs the same PacBio job quietly and with a thread that launches tiny torch kernels as fast as it can, and compares the
images byte for byte (and a window with the oracle).  usage: pb_noise_probe.py [lanes] [reads_per_lane] [rounds]"""
import hashlib
import sys
import threading
import time

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import torch

import jackalope_amd as ja

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = ja.synthetic_genome([40_000_000], seed=5)
words = ja.seed_words(9, 16 * T)
pb = {"custom_read_lengths": [600, 1500, 3000]}


def run(noise):
    stop = [False]
    count = [0]

    def hammer():
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        x = torch.zeros(64, device="cuda")
        with torch.cuda.stream(st):
            while not stop[0]:
                for _ in range(64):
                    x.add_(1.0)
                count[0] += 64
                st.synchronize()
    th = None
    s = ja.pacbio(g, None, T * per, n_threads=T, seed_words=words, _session=True, **pb)
    with s:
        if noise:
            th = threading.Thread(target=hammer)
            th.start()
            time.sleep(0.05)
        t0 = time.time()
        s.generate()
        dt = time.time() - t0
        stop[0] = True
        if th:
            th.join()
        img = s.fetch(0)
    return hashlib.sha256(img).hexdigest(), len(img), dt, count[0], img


h0, n0, dt, _, img0 = run(False)
print("quiet: %d bytes in %.3f s  %s" % (n0, dt, h0[:16]), flush=True)
bad = 0
for r in range(rounds):
    h, n, dt, c, img = run(True)
    same = (h == h0)
    print("noise round %d: %d bytes in %.3f s, %d foreign launches  %s  %s" % (r, n, dt, c, h[:16], "same" if same else "DIFFERENT"), flush=True)
    if not same:
        bad += 1
        a = np.frombuffer(img0, dtype=np.uint8); b = np.frombuffer(img, dtype=np.uint8)
        m = min(a.size, b.size)
        d = np.nonzero(a[:m] != b[:m])[0]
        print("   %d differing bytes, first at %d: quiet %r / noise %r" % (d.size, d[0], bytes(a[d[0] - 40:d[0] + 40]), bytes(b[d[0] - 40:d[0] + 40])), flush=True)
h, n, dt, _, _ = run(False)
print("quiet again: %s %s" % (h[:16], "same" if h == h0 else "DIFFERENT"))
print("RESULT: %d of %d noisy rounds differ" % (bad, rounds))
