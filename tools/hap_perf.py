"""Quick Illumina-on-haplotypes throughput probe (configs[2]-like, scaled): G Mbp in 4 chromosomes, H haplotypes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jackalope_amd as ja
from jackalope_amd.genome import HapSet

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 400.0
n_haps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
nc = int(os.environ.get("JK_NCHROM", "4"))
ref = ja.synthetic_genome([int(mbp * 1e6 / nc)] * nc, seed=3)
rng = np.random.default_rng(31)
cells = []
t = time.time()
for h in range(n_haps):          # vectorised synthetic tables: substitutions 1e-3/bp, 1-base insertions/deletions 1e-4/bp each
    row = []
    for seq in ref.seqs:
        n = seq.size
        k = int(n * float(os.environ.get("JK_MUT_RATE", "1.2e-3")))
        pos = np.unique(rng.integers(1, n - 2, size=k) & ~np.int64(3))     # spaced >= 4 apart: never adjacent
        kind = rng.choice(3, size=pos.size, p=[1 / 1.2, 0.1 / 1.2, 0.1 / 1.2])
        delta = np.where(kind == 1, 1, np.where(kind == 2, -1, 0))
        shift = np.concatenate([[0], np.cumsum(delta)[:-1]])
        lut = np.frombuffer(b"TCAG", dtype=np.uint8)
        sub = lut[rng.integers(0, 4, size=pos.size)]
        ins = lut[rng.integers(0, 4, size=pos.size)]
        nuc = []
        refb = seq[pos]
        for kd, rb, sb, ib in zip(kind.tolist(), refb.tolist(), sub.tolist(), ins.tolist()):
            nuc.append(chr(sb) if kd == 0 else (chr(rb) + chr(ib) if kd == 1 else ""))
        row.append({"chrom_size": int(n + delta.sum()), "old_pos": pos.tolist(), "new_pos": (pos + shift).tolist(), "nucleos": nuc})
    cells.append(row)
hs = HapSet(ref, cells)
print("tables built in %.1fs, %d mutations" % (time.time() - t, sum(len(c["new_pos"]) for r in cells for c in r)))
n_pairs = int(mbp * 1e6 * 30 / 300)
words = ja.seed_words(12345, hs.seed_budget(lanes))
t = time.time()
s = ja.illumina(hs, None, 2 * n_pairs, 150, True, n_threads=lanes, seed_words=words, _session=True)
print("open %.1fs" % (time.time() - t))
with s:
    for i in range(3):
        t = time.time(); s.generate(); dt = time.time() - t
        sizes, reads = s.sizes(); tm = s.timing_ms()
        print("pairs %d  wall %.1f ms  gen %.1f ms -> %.1f M pairs/s" % (reads // 2, dt * 1e3, tm["generate_kernel"], reads / 2 / dt / 1e6))
