"""End-to-end phases of one illumina() call on the headline workload (open, generate, write plain / BGZF)."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jackalope_amd as ja
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
t = time.time(); g = ja.synthetic_genome([100_000_000], seed=2); print("genome (numpy) %.2fs" % (time.time() - t))
t = time.time(); words = ja.seed_words(12345, 16 * lanes); print("seed words %.2fs" % (time.time() - t))
tmp = tempfile.mkdtemp(prefix="jk_e2e_")
try:
    for method, comp in (("plain", False), ("bgzip", True)):
        t0 = time.time()
        s = ja.illumina(g, os.path.join(tmp, method), 20_000_000, 150, True, n_threads=lanes, seed_words=words, _session=True,
                        compress=comp, overwrite=True)
        t1 = time.time(); s.generate(); t2 = time.time(); s.write(); t3 = time.time()
        sizes, reads = s.sizes(); s.close()
        fs = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp) if f.startswith(method))
        print("%s: open %.2fs  generate %.3fs  write %.2fs (%.2f GB FASTQ -> %.2f GB on disk)" % (method, t1 - t0, t2 - t1, t3 - t2, sum(sizes) / 1e9, fs / 1e9))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
