"""BASELINE configs[2] at full size on ONE MI355X: 3 Gbp reference (24 chromosomes of 125 Mbp, made on the device),
4 haplotypes (substitutions 1e-3/bp, 1-base insertions and deletions 1e-4/bp each), 30x Illumina PE150 = 300 M
pairs.  Default: the streaming session (two per-launch image slots, each launch's FASTQ handed to the sink -- the
null sink here -- as it completes).  --resident keeps the whole FASTQ image (about 200 GB) in HBM instead; its
hipMalloc alone then takes 4-6 s in the driver (tools/malloc_probe.hip: any allocation past the first ~64 GB of a
fresh process does), which is not set-up work of this library.  (Byte parity at this size: tests/test_gpu_full_size.py.)

    python tools/config3_full.py [--scale 1.0] [--lanes 2097152] [--resident]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jackalope_amd as ja
from jackalope_amd.genome import random_haplotypes_flat

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--lanes", type=int, default=1 << 21)
ap.add_argument("--haps", type=int, default=4)
ap.add_argument("--batch-gb", type=float, default=0, help="cap on the pool bytes per launch and read end (0 = the library decides: whole 2^18-lane launches, 12.3 GB here, when they fit)")
ap.add_argument("--resident", action="store_true")
ap.add_argument("--jobs", type=int, default=1, help="open, run and close that many sessions one after the other: the second finds the first one's "
                                                    "device buffers parked in the arena (JK_ARENA=0 to see it without)")
a = ap.parse_args()

t = time.time()
n_chroms, chrom_len = 24, int(125e6 * a.scale)
dev = ja.create_genome(n_chroms, chrom_len, 0, seed_words=ja.seed_words(3, 8))
ref = ja.RefGenome([dev.chrom(i) for i in range(n_chroms)], names=dev.names)
dev.close()
print("genome %.2f Gbp made on the device and fetched: %.1f s" % (n_chroms * chrom_len / 1e9, time.time() - t), flush=True)
t = time.time()
hs = random_haplotypes_flat(ref, a.haps, seed=31)
print("%d haplotypes, %d mutations: %.1f s" % (a.haps, int(hs.n_mut.sum()), time.time() - t), flush=True)
n_pairs = int(n_chroms * chrom_len * 30 / 300)
words = ja.seed_words(12345, a.lanes * (16 + 16 * a.haps) + 64)
for job in range(a.jobs):
    t = time.time()
    s = ja.illumina(hs, None, 2 * n_pairs, 150, True, n_threads=a.lanes, seed_words=words, max_batch_bytes=int(a.batch_gb * 1e9), _session=True,
                    stream_output=not a.resident)
    print("job %d: open (host planning, uploads, %d lanes): %.2f s; arena %s" % (job, a.lanes, time.time() - t, ja.arena_stats()), flush=True)
    with s:
        for rep in range(2 if a.jobs == 1 else 1):
            t = time.time(); (s.generate() if a.resident else s.run()); dt = time.time() - t
            sizes, reads = s.sizes(); tm = s.timing_ms()
            print("%s: %d pairs, %.1f + %.1f GB FASTQ in %.3f s (generator kernels %.3f s, %d launches) -> %.1f M pairs/s"
                  % ("generate, image resident" if a.resident else "run, every launch's FASTQ copied to pinned host buffers (null writer)", reads // 2, sizes[0] / 1e9, sizes[1] / 1e9, dt, tm["generate_kernel"] / 1e3, s.n_batches(), reads / 2 / dt / 1e6), flush=True)
        assert reads == 2 * n_pairs
        import torch
        print("HBM in use: %.1f GB" % ((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9))
