import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import jackalope_amd as ja, oracle_lib as O
from test_gpu_pacbio import hip
from helpers import fastq_records
seed, case = int(sys.argv[1]), int(sys.argv[2])
import fuzz_gpu
rng = np.random.default_rng([seed, case])
# replicate pacbio_case's draws up to the run
small = rng.random() < 0.35
sizes = [int(rng.integers(300, 6000)) if small else int(rng.integers(20_000, 300_000)) for _ in range(int(rng.choice([1, 2, 4])))]
g = ja.synthetic_genome(sizes, seed=int(rng.integers(0, 10 ** 6)))
T = int(rng.choice([1, 3, 64, 130]))
n = int(rng.integers(1, 6)) * T + int(rng.integers(0, 3))
if rng.random() < 0.2:
    n = int(rng.integers(20, 120)) * T
pb = {}
if rng.random() < 0.6:
    pb["custom_read_lengths"] = sorted(int(x) for x in rng.integers(100, max(min(sizes) // 3, 102) if not small or rng.random() < 0.5 else 2 * max(sizes), size=int(rng.integers(1, 5))))
if rng.random() < 0.3:
    pb["prob_dup"] = float(rng.choice([0.1, 0.5]))
if rng.random() < 0.3:
    pb["ins_prob"], pb["del_prob"], pb["sub_prob"] = float(rng.choice([0.05, 0.2])), float(rng.choice([0.02, 0.1])), float(rng.choice([0.005, 0.05]))
if rng.random() < 0.2:
    pb["max_passes"] = int(rng.choice([1, 4, 20]))
words = ja.seed_words(int(rng.integers(0, 2 ** 31)), 64 * T * 8 + 256)
print(sizes, T, n, pb)
h, reads, used = hip(ja, g, n, T, words, pb)
o, used_o, _ = O.pacbio_ref(g, pb, n_reads=n, n_threads=T, words=words)
rh, ro = h.split(b"\n"), o.split(b"\n")
for i in range(0, min(len(rh), len(ro)) - 1, 4):
    if rh[i:i+4] != ro[i:i+4]:
        k = i // 4
        print("first differing read", k, "of", len(ro) // 4)
        for j in (k - 1, k):
            a, b = rh[4*j:4*j+4], ro[4*j:4*j+4]
            print(" read", j, "hip id", a[0], "len seq", len(a[1]), "len qual", len(a[3]), "| orc id", b[0], "len seq", len(b[1]), "len qual", len(b[3]))
            if a[1] != b[1]:
                x = next((t for t in range(min(len(a[1]), len(b[1]))) if a[1][t] != b[1][t]), None)
                print("   seq differs at", x, a[1][max(0,(x or 0)-10):(x or 0)+10], b[1][max(0,(x or 0)-10):(x or 0)+10])
        break
else:
    print("identical" if h == o else "tail differs")
