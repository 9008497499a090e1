"""GPU, world_size = 2 over gloo, both ranks on the one device there is: the real N > 1 path through the HIP library.
Each rank opens its lane shard in O(own lanes) (seed-offset exchange, jackalope_amd.sharding.open_shard), generates,
all-gathers {reads, bytes} (sharding.exchange_counts), and writes its image at its offset of the shared files
(jk_session_write_shard).  The files must equal the single-process run, which the other tests pin to the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CASES = {
    # name: (chromosome sizes, n_haps, n_reads, T)
    "many_reads_per_lane": ([120_000, 60_000, 9_000], 3, 200_000, 1001),     # the speculated seed offsets hold
    "few_reads_per_lane": ([50_000, 20_000], 5, 3_000, 700),                  # haplotypes without reads: offsets are corrected
    "pacbio": ([400_000, 150_000], 0, 1_500, 333),                            # pacbio() on the reference genome, one read end
}
PB = {"custom_read_lengths": [700, 2500, 6000]}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(ja, case):
    from jackalope_amd.genome import random_haplotypes
    sizes, nh, n_reads, T = CASES[case]
    ref = ja.synthetic_genome(sizes, seed=61)
    if nh == 0:
        return ref, ja.seed_words(63, 16 * T + 64), n_reads, T
    hs = random_haplotypes(ref, nh, seed=62)
    words = ja.seed_words(63, hs.seed_budget(T) + 64)
    return hs, words, n_reads, T


def _worker(rank, world, port, tmpdir, case):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import jackalope_amd as ja
    from jackalope_amd.sharding import open_shard, exchange_counts
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hs, words, n_reads, T = _make(ja, case)
    prefix = os.path.join(tmpdir, "shared")
    opened = []

    def open_fn(lo, hi, off):
        opened.append(off)
        if case == "pacbio":
            return ja.pacbio(hs, prefix, n_reads, n_threads=T, seed_words=words, lane_begin=lo, lane_end=hi, seed_offset_words=off,
                             _session=True, **PB)
        return ja.illumina(hs, prefix, n_reads, 150, True, n_threads=T, seed_words=words, lane_begin=lo, lane_end=hi,
                           seed_offset_words=off, _session=True)
    if case == "pacbio":
        s = open_shard(open_fn, T, n_reads, 8)
    else:
        s = open_shard(open_fn, T, n_reads // 2, 8 + 16 * hs.n_haps())
    with s:
        s.generate()
        sizes, reads = s.sizes()
        offsets, totals = exchange_counts(reads, sizes)
        s.write_shard(offsets)
        seed_range = s.shard_seed_words()
    dist.barrier()
    with open(os.path.join(tmpdir, "rank%d.txt" % rank), "w") as fh:
        fh.write("%d %d %d %d %d %d" % (len(opened), seed_range[0], seed_range[1], totals[0], totals[1][0], totals[1][-1]))
    dist.destroy_process_group()


@pytest.mark.parametrize("case", sorted(CASES))
def test_two_ranks_on_one_gpu_write_the_single_process_files(ja, tmp_path, case):
    # older, LONGER files of the same names are in the way (overwrite = TRUE): the ranks open the shared files without
    # truncating them, so the rank that holds the last lane has to cut the old tail off
    for e in (1, 2):
        with open(tmp_path / ("shared_R%d.fq" % e), "wb") as fh:
            fh.write(b"@stale\nACGT\n+\n!!!!\n" * 8_000_000)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), case), nprocs=2, join=True)
    hs, words, n_reads, T = _make(ja, case)
    if case == "pacbio":
        with ja.pacbio(hs, None, n_reads, n_threads=T, seed_words=words, _session=True, **PB) as s:
            s.generate()
            r1 = r2 = s.fetch(0)
            used = s.seed_words_used()
        assert open(tmp_path / "shared_R1.fq", "rb").read() == r1
    else:
        with ja.illumina(hs, None, n_reads, 150, True, n_threads=T, seed_words=words, _session=True) as s:
            s.generate()
            r1, r2 = s.fetch(0), s.fetch(1)
            used = s.seed_words_used()
        assert open(tmp_path / "shared_R1.fq", "rb").read() == r1
        assert open(tmp_path / "shared_R2.fq", "rb").read() == r2
    info = [[int(x) for x in open(tmp_path / ("rank%d.txt" % r)).read().split()] for r in range(2)]
    assert info[0][1] == 8 * T and info[0][2] == info[1][1] and info[1][2] == used      # the ranks' seed ranges chain up
    assert info[0][3:] == info[1][3:] == [n_reads, len(r1), len(r2)]
    if case != "few_reads_per_lane":
        assert info[0][0] == 1 and info[1][0] == 1        # nobody had to open twice
    else:
        assert info[1][0] == 2                            # rank 1's speculated offset was wrong and was corrected
