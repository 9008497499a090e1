import sys, os, gzip, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import jackalope_amd as ja, oracle_lib as O
ref = ja.synthetic_genome([600_000], seed=24)
n_reads, T = 3000, 512
words = ja.seed_words(80, 16 * T)
o, _, _ = O.pacbio_ref(ref, {}, n_reads=n_reads, n_threads=T, words=words)
d = tempfile.mkdtemp()
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    a = os.path.join(d, "a%d" % it)
    ja.pacbio(ref, a, n_reads, n_threads=T, seed_words=words, max_batch_bytes=8 << 20)
    if open(a + "_R1.fq", "rb").read() != o:
        print(it, "plain differs"); bad += 1
    for devs in ([0, 0],):
        b = os.path.join(d, "b%d_%d" % (it, len(devs)))
        ja.pacbio(ref, b, n_reads, n_threads=T, seed_words=words, devices=devs, compress=3)
        raw = open(b + "_R1.fq.gz", "rb").read()
        try:
            x = gzip.decompress(raw)
        except Exception as e:
            print(it, devs, "decompress failed", e); bad += 1; continue
        if x != o:
            bad += 1
            i = next((k for k in range(min(len(x), len(o))) if x[k] != o[k]), None)
            print(it, devs, "differs: len", len(x), len(o), "first diff at", i)
            if i is not None:
                j = next((k for k in range(i, min(len(x), len(o))) if x[k] == o[k] and x[k:k+64] == o[k:k+64]), None)
                print("   equal again from", j, "| got", x[max(0,i-30):i+50], "| want", o[max(0,i-30):i+50])
print("bad", bad)
