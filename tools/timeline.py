"""Experiment (not product): per-wave start/end times of the Illumina generator, from a -DJK_TIMELINE build.

    hipcc ... -DJK_TIMELINE -o build_variants/libjk_timeline.so jackalope_amd/csrc/jk_api.hip
    JK_HIP_LIB=build_variants/libjk_timeline.so python tools/timeline.py [lanes] [pairs]

Prints, for the LAST generator launch of one generate(): the spread of wave start and end times (100 MHz wall
clock), per-XCD and per-CU finish times, and how much of the launch is tail."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import jackalope_amd as ja  # noqa: E402
from jackalope_amd import _abi  # noqa: E402

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2500000
torch.cuda.set_device(0)
genome = ja.synthetic_genome([100_000_000], seed=2)
words = ja.seed_words(12345, 16 * lanes)
sess = ja.illumina(genome, None, 2 * pairs, 150, True, n_threads=lanes, seed_words=words, device=0, _session=True)
for _ in range(3):
    sess.generate()
torch.cuda.synchronize()
print("timing", sess.timing_ms(), "batches", sess.n_batches())
L = _abi.lib()
n_waves = min(lanes // 64, 1 << 15)
buf = np.zeros(4 * n_waves, dtype=np.uint64)
L.jk_debug_timeline.argtypes = [C.c_void_p, C.c_uint64]
assert L.jk_debug_timeline(buf.ctypes.data, buf.size) == 0
t = buf.reshape(-1, 4)
t = t[t[:, 1] > 0]
t0, t1 = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
base = t0.min()
s = (t0 - base) / 100.0          # us
e = (t1 - base) / 100.0
hw, xcc = t[:, 2].astype(np.int64), t[:, 3].astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
print("waves", len(t), "span us", e.max())
print("start us: min %.1f p50 %.1f p99 %.1f max %.1f" % (s.min(), np.median(s), np.percentile(s, 99), s.max()))
print("end   us: min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (e.min(), np.percentile(e, 10), np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max()))
d = e - s
print("life  us: min %.1f p50 %.1f mean %.1f max %.1f" % (d.min(), np.median(d), d.mean(), d.max()))
for x in range(8):
    m = xcc == x
    if m.any():
        print("xcc %d: waves %5d start max %.1f  end mean %.1f max %.1f  life mean %.1f" % (x, m.sum(), s[m].max(), e[m].mean(), e[m].max(), d[m].mean()))
key = xcc * 1000 + se * 100 + sh * 50 + cu
uniq = np.unique(key)
print("distinct (xcc,se,sh,cu):", len(uniq))
per = np.array([(k, (key == k).sum(), d[key == k].mean(), e[key == k].max()) for k in uniq])
print("waves per CU: min %d max %d" % (per[:, 1].min(), per[:, 1].max()))
order = np.argsort(per[:, 3])
print("slowest CUs (key, waves, mean life, end):", per[order[-5:]].tolist())
print("fastest CUs:", per[order[:5]].tolist())
widx = np.arange(len(buf) // 4)[buf.reshape(-1, 4)[:, 1] > 0] & 15
print("life by wave index in its workgroup:", " ".join("%d:%.0f" % (w, d[widx == w].mean()) for w in range(16)))
slot = hw & 0xf
print("life by HW wave slot:", " ".join("%d:%.0f(%d)" % (w, d[slot == w].mean(), (slot == w).sum()) for w in range(16) if (slot == w).any()))
for sm in range(4):
    m = simd == sm
    print("simd %d: waves %d life mean %.1f" % (sm, m.sum(), d[m].mean()))
sess.close()
