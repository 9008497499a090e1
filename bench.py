#!/usr/bin/env python3
"""bench.py -- headline benchmark of the read-generation hot path on MI355X.

Metric (BASELINE.json): M paired reads/sec for Illumina PE150 at 30x of a 100 Mbp synthetic
reference (configs[1]), per GPU count, plus the achieved fraction of the HBM roofline and the CPU
path (the oracle restatement of the reference) timed on this box's host cores in the same run.

A "step" is one pass of the hot path over the whole 30x job of one GPU: 10 M read pairs generated
by `--lanes` independent generator streams, pool-compacted into the two lane-major FASTQ images,
all resident in HBM (genome, tables, seeds and quotas are uploaded before the timed region; nothing
is copied to the host inside it).  With N GPUs the job is N x 10 M pairs over N x lanes lanes and
rank r generates lane block r (weak scaling; no data-path collective -- the only exchange is the
all-gather of per-rank pair/byte counts after the timed region).

  python bench.py [--gpus N --steps K --warmup W] [--lanes L] [--pairs P] [--genome-mbp G]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

XDEV = "cuda"
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Lanes per GPU: a generator launch takes 224 workgroups of 1024 lanes (the other 32 CUs run the compaction of the
# launch before), the first launch of a step all 256; four launches per step, 10.5 pairs per lane on
# the headline workload.
DEFAULT_LANES = (256 + 3 * 224) * 1024


def cpu_baseline(genome, prof1, prof2, read_length, sample_pairs, cores):
    """Time the CPU oracle (port of the reference's path) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import jackalope_amd as ja
    O.lib().orc_set_threads(cores)
    lanes = max(cores * 8, 8)
    words = ja.seed_words(12345, 16 * lanes)
    t0 = time.perf_counter()
    tb = {}
    O.illumina_ref(genome, paired=True, n_reads=2 * sample_pairs, prob_dup=0.02, n_threads=lanes,
                   read_pool_size=1000, shape=16.0, scale=25.0, fmin=read_length, fmax=2 ** 32 - 1,
                   prof1=prof1, prof2=prof2, ins1=0.00009, del1=0.00011, ins2=0.00015, del2=0.00023,
                   words=words, discard=True, thread_bytes=tb)
    dt = time.perf_counter() - t0
    pairs = sample_pairs
    return {"value": round(pairs / dt / 1e6, 6), "unit": "M paired reads/sec", "cores": cores, "kind": "port",
            "sample": "%d pairs of the same 100 Mbp PE150 workload on %d lanes, oracle/jk_oracle.cpp (CPU restatement "
                      "of the reference path) with OpenMP on %d threads, null sink: %d FASTQ bytes formatted and "
                      "dropped (%.1f s)" % (pairs, lanes, cores, int(tb[0].sum() + tb[1].sum()), dt)}


def pmc_traffic(pairs_per_launch):
    """HBM bytes per generator launch from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
    collected in separate passes, KB units, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950), if they were taken at this launch size."""
    path = os.path.join(ROOT, "profiles", "r03_pmc_generator.json")
    try:
        d = json.load(open(path))
        if abs(d["pairs_per_launch"] - pairs_per_launch) > 0.01 * pairs_per_launch:
            return None
        return int((2.0 * d["fetch_size_kb"] + d["write_size_kb"]) * 1024)
    except (OSError, KeyError, ValueError):
        return None


def pmc_traffic_pacbio(reads_per_launch):
    """The same for a PacBio launch (plan kernel + emit kernel), from profiles/r03_pmc_pacbio.json: taken on the bench
    workload, whose launches hold 1.5 M reads; null at another launch size."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_pacbio.json")))
        if abs(reads_per_launch - 1.5e6) > 0.02 * 1.5e6:
            return None
        return int(sum(v["traffic_bytes_per_launch (2*FETCH + WRITE, KB units)"] for v in d.values()))
    except (OSError, KeyError, ValueError, TypeError):
        return None


def bgzf_main(a):
    """Secondary line: the compressed sink.  The R1 FASTQ image of the headline workload (configs[1], 10 M pairs,
    3.3 GB) is BGZF-compressed where it lies in HBM (jk_bgzf_deflate).  A step = one pass over that image."""
    import ctypes as C
    import torch
    import jackalope_amd as ja
    from jackalope_amd import _abi
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("the BGZF line is single-GPU in this round")
    torch.cuda.set_device(local_rank)
    genome = ja.synthetic_genome([int(a.genome_mbp * 1e6)], seed=2)
    words = ja.seed_words(12345, 16 * a.lanes)
    sess = ja.illumina(genome, None, 2 * a.pairs, 150, True, n_threads=a.lanes, seed_words=words, device=local_rank, _session=True)
    sess.generate()
    sizes, _ = sess.sizes()
    n = int(sizes[0])
    L = _abi.lib()
    cap = int(L.jk_bgzf_bound(n))
    dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    src = sess.device_ptr(0)
    out_bytes, ms = C.c_uint64(), C.c_double()

    def step():
        _abi.check(L.jk_bgzf_deflate(local_rank, src, n, dst.data_ptr(), cap, C.byref(out_bytes), C.byref(ms)))
        return ms.value
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev_ms = 0.0
    for _ in range(a.steps):
        dev_ms += step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    comp = int(out_bytes.value)
    n_blocks = (n + 0xff00 - 1) // 0xff00
    n_launch = (n_blocks + 8191) // 8192
    alg = n + comp                                   # read the image once, write the compressed image once
    kern_s = dev_ms / a.steps / 1e3
    out = {"metric": "GB/s of FASTQ into BGZF (device sink)", "value": round(n * a.steps / elapsed / 1e9, 2), "unit": "GB/s",
           "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "R1 FASTQ image of configs[1] (%d pairs PE150, %d bytes) compressed in HBM" % (a.pairs, n),
                      "bgzf_blocks": n_blocks, "compressed_bytes": comp, "ratio": round(comp / n, 4)},
           "roofline": {"bound": "hbm", "achieved": round(alg / kern_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / kern_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                        "kernel": "bgzf_deflate_kernel (+ scan, gather)", "launches_per_step": n_launch,
                        "kernel_ms": round(kern_s * 1e3 / n_launch, 3),
                        "note": "achieved = (input + compressed bytes) / device time of all launches of a step; the "
                                "slot scratch adds one write and one read of the compressed bytes on top"}}
    if not a.no_cpu_baseline:
        # what the reference does on the host (bgzf_write: zlib deflate of 0xff00-byte blocks), level 6, all cores
        import zlib
        from concurrent.futures import ThreadPoolExecutor
        cores = min(os.cpu_count() or 1, 64)
        sample_n = min(n, 0xff00 * 4096 * 4)
        plain = sess.fetch(0)[:sample_n]

        def one(off):
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            blk = plain[off:off + 0xff00]
            zlib.crc32(blk)
            return len(co.compress(blk) + co.flush()) + 26
        t1 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            tot = sum(ex.map(one, range(0, sample_n, 0xff00)))
        dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(sample_n / dt / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "port",
                               "sample": "first %d bytes of the same image, zlib level 6 raw deflate per 0xff00-byte block "
                                         "(what bgzf_write does) on %d threads, ratio %.4f (%.1f s)" % (sample_n, cores, tot / sample_n, dt)}
    print(json.dumps(out))
    sess.close()


def create_genome_main(a):
    """Secondary line: create_genome() on the device -- BASELINE configs[2]'s 3 Gbp reference (24 chromosomes of
    125 Mbp, equal base frequencies) made in HBM.  A step = one whole genome."""
    import torch
    import jackalope_amd as ja
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("the create_genome line is single-GPU in this round")
    torch.cuda.set_device(local_rank)
    n_chroms, chrom_len = 24, int(125e6)
    words = ja.seed_words(12345, 8)

    def step():
        g = ja.create_genome(n_chroms, chrom_len, 0, n_threads=1, seed_words=words, device=local_rank)
        ms = g.kernel_ms()
        g.close()
        return ms
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev_ms = 0.0
    for _ in range(a.steps):
        dev_ms += step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    bases = n_chroms * chrom_len
    kern_s = dev_ms / a.steps / 1e3
    out = {"metric": "Gbases/sec of create_genome (3 Gbp, 24 chromosomes)", "value": round(bases * a.steps / elapsed / 1e9, 2),
           "unit": "Gbases/sec", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": "create_genome(24, 125e6, len_sd = 0, pi_tcag = 1/4 each, n_threads = 1): the reference "
                                  "genome of configs[2..3]", "bases": bases},
           "roofline": {"bound": "hbm", "achieved": round(bases / kern_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(bases / kern_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": None, "kernel": "create_genome_kernel",
                        "launches_per_step": 1, "kernel_ms": round(kern_s * 1e3, 3),
                        "note": "1 byte written per base; integer-ALU bound: 2 pcg64 outputs per base (%.3g outputs/s "
                                "against the measured 1.45e12/s ceiling)" % (2 * bases / kern_s)}}
    if not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        cores = min(os.cpu_count() or 1, 64)
        sample_len = int(400e6)
        t1 = time.perf_counter()
        O.create_genome(1, float(sample_len), 0.0, [0.25] * 4, 1, words)
        dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(sample_len / dt / 1e9, 4), "unit": "Gbases/sec", "cores": 1, "kind": "port",
                               "sample": "one %d-base chromosome by the oracle on 1 thread, as the reference runs with its default "
                                         "n_threads = 1 (one engine is a sequential chain; %d host cores idle) (%.1f s)" % (sample_len, cores - 1, dt)}
    print(json.dumps(out))


def read_fasta_main(a):
    """Secondary line: read_fasta() of an uncompressed FASTA (24 chromosomes, 80 columns) made by this run in a
    temporary directory.  `value` is end to end (host file read + upload + device packing); the roofline is the
    device packing alone.  --genome-mbp sets the size (default 1000 Mbp)."""
    import shutil
    import tempfile
    import numpy as np
    import torch
    import jackalope_amd as ja
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    mbp = a.genome_mbp if a.genome_mbp != 100.0 else 1000.0
    n_chroms = 24
    chrom_len = int(mbp * 1e6 / n_chroms) // 80 * 80
    tmp = tempfile.mkdtemp(prefix="jk_bench_fa_")
    try:
        fn = os.path.join(tmp, "ref.fa")
        g0 = ja.create_genome(n_chroms, chrom_len, 0, seed_words=ja.seed_words(5, 8), device=local_rank)
        with open(fn, "wb") as f:
            for i in range(n_chroms):
                f.write(b">chrom%d\n" % i)
                rows = g0.chrom(i).reshape(-1, 80)
                f.write(np.concatenate([rows, np.full((rows.shape[0], 1), 10, dtype=np.uint8)], axis=1).tobytes())
        g0.close()
        file_bytes = os.path.getsize(fn)
        bases = n_chroms * chrom_len

        def step():
            g = ja.read_fasta(fn, device=local_rank)
            ms = g.kernel_ms()
            g.close()
            return ms
        for _ in range(a.warmup):
            step()
        t0 = time.perf_counter()
        dev_ms = 0.0
        for _ in range(a.steps):
            dev_ms += step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kern_s = dev_ms / a.steps / 1e3
        alg = file_bytes + bases                   # text read once, bases written once
        out = {"metric": "Gbases/sec of read_fasta (uncompressed, end to end)", "value": round(bases * a.steps / elapsed / 1e9, 3),
               "unit": "Gbases/sec", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "read_fasta of a %d-byte FASTA (24 chromosomes, 80 columns, page cache)" % file_bytes, "bases": bases},
               "roofline": {"bound": "hbm", "achieved": round(alg / kern_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(alg / kern_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                            "kernel": "fasta_count_kernel + scan + fasta_pack_kernel", "launches_per_step": 1,
                            "kernel_ms": round(kern_s * 1e3, 3),
                            "note": "device part only; the count pass re-reads the text, so HBM traffic is about 2x text + bases"}}
        if not a.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            t1 = time.perf_counter()
            O.read_fasta([fn])
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": round(bases / dt / 1e9, 4), "unit": "Gbases/sec", "cores": 1, "kind": "port",
                                   "sample": "the same file through the oracle's restatement of read_fasta_noind (single-threaded, "
                                             "as the reference is) (%.1f s)" % dt}
        print(json.dumps(out))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def dist_setup(a):
    """One process per GPU: rank/world from the launcher's environment, RCCL (backend "nccl") for the count exchange."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with that many ranks" % a.gpus)
        a.gpus = world
    # Rehearsal of the N > 1 code path on a box with one GPU: JK_BENCH_ONE_DEVICE=1 puts every rank on device 0 and
    # JK_BENCH_BACKEND=gloo replaces RCCL (which wants a device per rank) for the few integers the ranks exchange.
    if os.environ.get("JK_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    global XDEV
    backend = os.environ.get("JK_BENCH_BACKEND", "nccl")
    XDEV = "cuda" if backend == "nccl" else "cpu"          # where the exchanged integers live
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("JK_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world, use_dist


def timed_steps(a, sess, use_dist, pipelined=False):
    """W warm-up passes, then K passes between barrier + synchronize on both sides; MAX over ranks of the elapsed time."""
    import torch
    import torch.distributed as dist

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
    for _ in range(a.warmup):
        sess.generate()
    sync()
    t0 = time.perf_counter()
    gen_ms = all_ms = 0.0
    global STEP_MS
    STEP_MS = []
    if pipelined:
        # K steps, each a whole job (every lane's reads generated and assembled into the FASTQ image in HBM), queued so
        # that two are in flight: the next step's first generator launch runs beside this step's last compaction -- the
        # way job follows job in a caller with more than one (sep_files haplotypes, call after call).  The per-launch
        # kernel time comes from the last step's HIP events.
        t1 = time.perf_counter()
        for k in range(a.steps):
            sess.generate_async()
            if k >= 1:
                sess.wait()
                t2 = time.perf_counter(); STEP_MS.append((t2 - t1) * 1e3); t1 = t2
        sess.wait()
        STEP_MS.append((time.perf_counter() - t1) * 1e3)
        tm = sess.timing_ms()
        gen_ms, all_ms = tm["generate_kernel"] * a.steps, tm["total"] * a.steps
    else:
        for _ in range(a.steps):
            t1 = time.perf_counter()
            sess.generate()                      # blocks until the step's stream work is done
            STEP_MS.append((time.perf_counter() - t1) * 1e3)
            tm = sess.timing_ms()                # HIP events on the stream the kernels run on
            gen_ms += tm["generate_kernel"]
            all_ms += tm["total"]
    sync()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=XDEV)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return float(tmax.item()), gen_ms, all_ms


STEP_MS = []


def step_spread():
    """min / median / max wall time of this rank's timed steps (each generate() returns when its step's device work is
    done): box-to-box and step-to-step spread belongs in the record next to the mean the contract asks for."""
    v = sorted(STEP_MS)
    if not v:
        return None
    return {"min": round(v[0], 3), "median": round(v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2]), 3),
            "max": round(v[-1], 3)}


def measured_copy_gbs():
    """The second HBM denominator: bytes read + written per second by a plain device-to-device
    copy of 2 GiB on this box."""
    import torch
    n = 2 << 30
    src = torch.empty(n, dtype=torch.uint8, device="cuda")
    dst = torch.empty(n, dtype=torch.uint8, device="cuda")
    src.fill_(7)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * 10 / (e0.elapsed_time(e1) / 1e3) / 1e9


def d2h_inclusive(open_stream, steps, compress):
    """Whole-job rate with the FASTQ (or its device-made BGZF form) copied to host memory as it is generated: a
    streaming session with a null sink (pinned double-buffered D2H overlapped with the next launch; nothing is written)."""
    s = open_stream(compress)
    with s:
        s.run()                                  # warm-up: pinned buffers, BGZF tables
        t0 = time.perf_counter()
        for _ in range(steps):
            s.run()
        dt = time.perf_counter() - t0
        _, reads = s.sizes()
    return reads * steps / dt


def pacbio_main(a):
    """Secondary line: PacBio reads (uniform 5-15 kb custom lengths, mean 10 kb) at 20x of a synthetic genome:
    BASELINE configs[4] at full size per GPU (3 Gbp, 6 M reads, 120 GB of FASTQ kept in HBM) unless --genome-mbp says
    otherwise; N GPUs take N such jobs' lanes (weak scaling), lane blocks via the seed-offset exchange."""
    rank, local_rank, world, use_dist = dist_setup(a)
    import torch
    import torch.distributed as dist
    import jackalope_amd as ja
    from jackalope_amd.sharding import open_shard, exchange_counts
    mbp = a.genome_mbp if a.genome_mbp != 100.0 else 3000.0
    genome = ja.synthetic_genome([int(mbp * 1e6)], seed=3)
    n_reads = int(mbp * 1e6 * 20 / 10000) * world
    # 2^21 lanes per GPU (~3 reads per lane): a launch of 2^18 lanes = 4096 waves of 64 lanes, four on every SIMD, which is what
    # the plan kernel wants; with fewer lanes the library spreads them over more waves (2^20 lanes: 38.7, 2^16: 26 M reads/s)
    lanes = (a.lanes if a.lanes != DEFAULT_LANES else (1 << 21)) * world
    lens = list(range(5000, 15001, 500))
    words = ja.seed_words(12345, 16 * lanes)
    sess = open_shard(lambda lo, hi, off: ja.pacbio(genome, None, n_reads, n_threads=lanes, seed_words=words, custom_read_lengths=lens,
                                                    device=local_rank, lane_begin=lo, lane_end=hi, seed_offset_words=off, _session=True),
                      lanes, n_reads, 8, device=XDEV if use_dist else None)
    # steps two in flight, as on the Illumina lines: the first plan kernel of step k + 1 runs beside the last emit kernel of step k
    elapsed, gen_ms, all_ms = timed_steps(a, sess, use_dist, pipelined=not a.sync_steps)
    spread = step_spread()
    sync_elapsed = None
    if not a.sync_steps and not a.no_extras:        # the same K steps one at a time (host sync after every step), for the record
        sync_elapsed, _, _ = timed_steps(a, sess, use_dist)
    sizes, reads = sess.sizes()
    offsets, (total_reads, total_bytes) = exchange_counts(reads, sizes, device=XDEV)
    if rank == 0:
        n_launch = max(sess.n_batches(), 1)
        alg = (sizes[0] + sizes[0] // 2) / n_launch                  # FASTQ bytes (~2 per base) + 1 reference byte per base
        # one launch = plan kernel + lane scan + emit kernel.  One step at a time: the step's HIP events.  Two steps in flight: a
        # step's events span more than its share of the device (its head runs beside the step before, its tail beside the next):
        # the steady-state time per step is the wall time between the barriers over the steps
        kern_s = (all_ms / a.steps / 1e3 if a.sync_steps else elapsed / a.steps) / n_launch
        out = {"metric": "M PacBio reads/sec (mean 10 kb, 20x)", "value": round(total_reads * a.steps / elapsed / 1e6, 3),
               "unit": "M reads/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": {"workload": "configs[4]: %g Mbp synthetic ref, PacBio defaults, custom read lengths uniform "
                                      "5-15 kb (mean 10 kb), 20x per GPU" % mbp, "reads_per_gpu": n_reads // world, "lanes_per_gpu": lanes // world,
                          "parallelism": "lanes sharded over %d GPU(s), no data-path collective" % world},
               "gbases_per_sec": round(total_bytes[0] / 2 * a.steps / elapsed / 1e9, 2), "step_ms": spread,
               "steps_mode": "one at a time" if a.sync_steps else "pipelined: two steps in flight, the next step's first plan kernel beside this step's last emit kernel",
               "roofline": {"bound": "hbm", "achieved": round(alg / kern_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(alg / kern_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": pmc_traffic_pacbio(reads / n_launch), "kernel": "pb_plan_kernel + pb_emit_kernel",
                            "launches_per_step": n_launch, "kernel_ms": round(kern_s * 1e3, 3), "plan_kernel_ms": round(gen_ms / a.steps / n_launch, 3),
                            "note": "a launch is a plan kernel (pass 1 of its reads: 64 positions per LCG jump-ahead step) and an emit kernel "
                                    "(one wave per read, text straight into the image); kernel_ms = time of a step / launches; "
                                    "plan_kernel_ms = time with a plan kernel running / launches (it shares the device with the previous "
                                    "launch's emit kernel)"}}
        if sync_elapsed is not None:
            out["value_one_step_at_a_time"] = round(total_reads * a.steps / sync_elapsed / 1e6, 3)
        if not a.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            cores = min(os.cpu_count() or 1, 64)
            O.lib().orc_set_threads(cores)
            cl = cores * 4
            sample = 3000 * cl
            t1 = time.perf_counter()
            O.pacbio_ref(genome, {"custom_read_lengths": lens}, n_reads=sample, n_threads=cl, words=ja.seed_words(1, 16 * cl), discard=True)
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": round(sample / dt / 1e6, 6), "unit": "M reads/sec", "cores": cores, "kind": "port",
                                   "sample": "%d reads of the same workload on %d lanes, oracle with OpenMP, null sink (%.1f s)" % (sample, cl, dt)}
        print(json.dumps(out))
    sess.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def hap_main(a):
    """Secondary line: BASELINE configs[3]'s share of one GPU -- 3 Gbp (24 x 125 Mbp, made on the device), 8 haplotypes
    (28.7 M mutations), 37.5 M PE150 pairs per GPU on --lanes lanes; N GPUs take N shares (at N = 8: the whole
    configs[3] job), lane blocks via the seed-offset exchange (RCCL all-gather of two integers per rank)."""
    rank, local_rank, world, use_dist = dist_setup(a)
    import torch
    import torch.distributed as dist
    import jackalope_amd as ja
    from jackalope_amd.genome import random_haplotypes_flat
    from jackalope_amd.sharding import open_shard, exchange_counts
    n_chroms, chrom_len, n_haps = 24, int(125e6 * (a.genome_mbp / 100.0 if a.genome_mbp != 100.0 else 1.0)), 8
    dev = ja.create_genome(n_chroms, chrom_len, 0, seed_words=ja.seed_words(3, 8), device=local_rank)
    ref = ja.RefGenome([dev.chrom(i) for i in range(n_chroms)], names=dev.names)
    dev.close()
    hs = random_haplotypes_flat(ref, n_haps, seed=31)
    pairs_per_gpu = n_chroms * chrom_len * 30 // 300 // 8
    if not a.sync_steps:
        # steps two in flight, as on the headline line: every launch leaves an eighth of the CUs to the compaction before it
        os.environ.setdefault("JK_FIRST_LAUNCH_FULL", "0")
        if a.lanes == DEFAULT_LANES:
            a.lanes = 4 * 224 * 1024
    lanes_per_gpu = a.lanes
    lanes, n_reads = lanes_per_gpu * world, 2 * pairs_per_gpu * world
    words = ja.seed_words(12345, hs.seed_budget(lanes))
    t0 = time.perf_counter()
    sess = open_shard(lambda lo, hi, off: ja.illumina(hs, None, n_reads, 150, True, n_threads=lanes, seed_words=words, device=local_rank,
                                                      lane_begin=lo, lane_end=hi, seed_offset_words=off, _session=True),
                      lanes, n_reads // 2, 8 + 16 * n_haps, device=XDEV if use_dist else None)
    open_s = time.perf_counter() - t0
    elapsed, gen_ms, all_ms = timed_steps(a, sess, use_dist, pipelined=not a.sync_steps)
    sizes, reads = sess.sizes()
    offsets, (total_reads, total_bytes) = exchange_counts(reads, sizes, device=XDEV)
    if rank == 0:
        n_launch = max(sess.n_batches(), 1)
        alg = (sum(sizes) + 300 * (reads // 2)) / n_launch
        kern_s = gen_ms / a.steps / 1e3 / n_launch
        print(json.dumps({
            "metric": "M paired reads/sec (PE150, 3 Gbp x 8 haplotypes)", "value": round(total_reads / 2 * a.steps / elapsed / 1e6, 3),
            "unit": "M paired reads/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic", "step_ms": step_spread(),
            "config": {"workload": "configs[3] share per GPU: %.3g Gbp reference, 8 haplotypes (%d mutations), 30x PE150 / 8"
                                   % (n_chroms * chrom_len / 1e9, int(hs.n_mut.sum())), "pairs_per_gpu": pairs_per_gpu,
                       "lanes_per_gpu": lanes_per_gpu, "open_seconds": round(open_s, 2),
                       "parallelism": "lanes sharded over %d GPU(s); seed-offset and count exchanges only" % world},
            "roofline": {"bound": "hbm", "achieved": round(alg / kern_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / kern_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": None, "kernel": "illumina_kernel<LDS,2,1024,hap>",
                         "launches_per_step": n_launch, "kernel_ms": round(kern_s * 1e3, 3)}}))
    sess.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--lanes", type=int, default=DEFAULT_LANES, help="generator lanes per GPU (default: four launches, the first of 256 x 1024 lanes, three of 224 x 1024)")
    ap.add_argument("--pairs", type=int, default=10_000_000, help="read pairs per GPU (30x of 100 Mbp at PE150)")
    ap.add_argument("--genome-mbp", type=float, default=100.0)
    ap.add_argument("--cpu-sample-pairs", type=int, default=0, help="0 = choose for about 15 s of CPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the D2H-inclusive and copy-bandwidth measurements")
    ap.add_argument("--sync-steps", action="store_true", help="Illumina lines: one step at a time (host sync after every step) instead of two in flight")
    ap.add_argument("--workload", choices=["illumina", "hap", "pacbio", "bgzf", "create_genome", "read_fasta"], default="illumina",
                    help="illumina = the headline metric (BASELINE configs[1]); hap = configs[3]'s share of one GPU; pacbio = "
                         "configs[4]; bgzf = the device-side compressed sink on the headline workload's FASTQ")
    a = ap.parse_args()
    if a.workload == "pacbio":
        return pacbio_main(a)
    if a.workload == "hap":
        return hap_main(a)
    if a.workload == "bgzf":
        return bgzf_main(a)
    if a.workload == "create_genome":
        return create_genome_main(a)
    if a.workload == "read_fasta":
        return read_fasta_main(a)

    rank, local_rank, world, use_dist = dist_setup(a)
    import torch
    import torch.distributed as dist
    import jackalope_amd as ja
    from jackalope_amd.sharding import open_shard, exchange_counts

    read_length = 150
    genome = ja.synthetic_genome([int(a.genome_mbp * 1e6)], seed=2)
    if not a.sync_steps:
        # steps two in flight: every launch, the first of a step included, leaves an eighth of the CUs to the compaction of
        # the launch before it (the previous step's last one): four launches of 224 x 1024 lanes
        os.environ.setdefault("JK_FIRST_LAUNCH_FULL", "0")
        if a.lanes == DEFAULT_LANES:
            a.lanes = 4 * 224 * 1024
    total_lanes = a.lanes * world
    n_reads = 2 * a.pairs * world
    words = ja.seed_words(12345, 16 * total_lanes)
    sess = open_shard(lambda lo, hi, off: ja.illumina(genome, None, n_reads, read_length, True, n_threads=total_lanes, seed_words=words,
                                                      device=local_rank, lane_begin=lo, lane_end=hi, seed_offset_words=off, _session=True),
                      total_lanes, n_reads // 2, 8, device=XDEV if use_dist else None)
    elapsed, gen_ms, all_ms = timed_steps(a, sess, use_dist, pipelined=not a.sync_steps)
    spread = step_spread()
    sync_elapsed = None
    if not a.sync_steps and not a.no_extras:        # the same K steps one at a time (host sync after every step), for the record
        sync_elapsed, _, _ = timed_steps(a, sess, use_dist)

    sizes, reads_made = sess.sizes()
    pairs_rank = reads_made // 2
    fastq_bytes = sum(sizes)

    # the path's only exchanges: the seed-offset all-gather inside open_shard and, here, per-rank {reads, bytes R1,
    # bytes R2} all-gathered over RCCL -> file offsets and totals
    offsets, (total_reads, total_bytes) = exchange_counts(reads_made, sizes, device=XDEV)
    total_pairs = total_reads // 2

    if rank == 0:
        value = total_pairs * a.steps / elapsed / 1e6
        # roofline of the dominant kernel (the generator): algorithmic bytes per launch = FASTQ bytes it
        # emits + 300 reference bytes per pair, over its average HIP-event duration
        n_launch = max(sess.n_batches(), 1)
        alg_bytes = (fastq_bytes + 300 * pairs_rank) / n_launch          # per launch
        kern_s = gen_ms / a.steps / 1e3 / n_launch                       # average launch duration
        achieved = alg_bytes / kern_s / 1e9
        traffic = pmc_traffic(pairs_rank / n_launch)
        out = {
            "metric": "M paired reads/sec (PE150, 30x of 100 Mbp)", "value": round(value, 3),
            "unit": "M paired reads/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[1]: %g Mbp synthetic ref (iid TCAG, seed 2), 1 haplotype, 30x Illumina "
                                   "PE150, HiSeq 2500 profile, default indel/dup probabilities" % a.genome_mbp,
                       "pairs_per_gpu": a.pairs, "lanes_per_gpu": a.lanes, "read_length": read_length,
                       "parallelism": "lanes sharded over %d GPU(s), no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": "illumina_kernel<LDS,2,1024,ref>", "launches_per_step": n_launch,
                         "kernel_ms": round(kern_s * 1e3, 3),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "algorithmic_bytes_per_pair": round(alg_bytes * n_launch / max(pairs_rank, 1), 2),
                         "note": "integer-ALU bound: ~1206 pcg64 steps (128-bit multiply) per pair, "
                                 "%.3g draws/s against a measured pure-pcg ceiling of 1.45e12 draws/s/GPU; "
                                 "traffic = (2*FETCH_SIZE + WRITE_SIZE) of the committed rocprofv3 --pmc passes "
                                 "(profiles/), null when they were taken at another launch size"
                                 % (1206.0 * pairs_rank / n_launch / kern_s)},
            "device_ms_per_step": round(all_ms / a.steps, 3), "step_ms": spread,
            "steps_mode": "one at a time" if a.sync_steps else "pipelined: two steps in flight, the next step's first launch beside this step's last compaction",
        }
        if sync_elapsed is not None:
            out["value_one_step_at_a_time"] = round(total_pairs * a.steps / sync_elapsed / 1e6, 3)
        # Since round 2 a launch takes 7/8 of the CUs and the compaction of the launch before runs on the rest, beside
        # it: per launch the generator kernel is slower than on the whole chip (roofline.kernel_ms), the step is faster.
        # The same algorithmic bytes over the whole step:
        step_gbs = (fastq_bytes + 300 * pairs_rank) / (elapsed / a.steps) / 1e9
        out["roofline_whole_step"] = {"achieved": round(step_gbs, 2), "unit": "GB/s", "frac": round(step_gbs / HBM_PEAK_GBS, 5),
                                      "note": "algorithmic bytes of a step / ms_per_step (generator launches on 224 CUs with the "
                                              "scan + compaction of the previous launch on the other 32, plus the last "
                                              "launch's compaction)"}
    sess.close()
    if rank == 0 and world == 1 and not a.no_extras:
        # the second HBM denominator and the D2H-inclusive rates (never `value`)
        copy_gbs = measured_copy_gbs()
        out["roofline"]["peak_measured_copy"] = round(copy_gbs, 1)
        out["roofline"]["frac_of_measured_copy"] = round(out["roofline"]["achieved"] / copy_gbs, 5)

        def open_stream(compress):
            return ja.illumina(genome, None, n_reads, read_length, True, n_threads=total_lanes, seed_words=words, device=local_rank,
                               compress=(compress or False), _session=True, stream_output=True)
        out["value_incl_d2h"] = round(d2h_inclusive(open_stream, 3, 0) / 2 / 1e6, 3)
        out["value_incl_d2h_bgzf"] = round(d2h_inclusive(open_stream, 3, 6) / 2 / 1e6, 3)
        out["value_incl_d2h_note"] = ("M pairs/s with every launch's FASTQ copied to pinned host memory while the next launch runs "
                                      "(streaming session, null sink); _bgzf: compressed on the device first (LZ77 matches against the previous record + two Huffman codes), 0.33x the bytes")
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:            # the CPU baseline is reported at N = 1 only
            cores = min(os.cpu_count() or 1, 64)
            sample = a.cpu_sample_pairs or min(200_000 * cores, 12_000_000)     # about 15-20 s of CPU work
            p1, p2 = ja.read_profile(None, None, read_length, 1), ja.read_profile(None, None, read_length, 2)
            out["cpu_baseline"] = cpu_baseline(genome, p1, p2, read_length, sample, cores)
            one = cpu_baseline(genome, p1, p2, read_length, 400_000, 1)
            out["cpu_baseline_1thread"] = one
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
