"""CPU-only, world_size = 2 over gloo: the N>1 path.  Each rank generates its lane block (here with the
oracle standing in for the GPU, which is absent), exchanges counts exactly as bench.py / a multi-GPU
host does (jackalope_amd.sharding), writes its image at its offset of a shared file, and the result must
equal the single-process output."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir, T, n_reads):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import jackalope_amd as ja
    import oracle_lib as O
    from jackalope_amd.sharding import lane_block, exchange_counts
    from helpers import job, run_oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = ja.synthetic_genome([60_000, 7_000], seed=3)
    p1, p2 = ja.read_profile(None, None, 150, 1), ja.read_profile(None, None, 150, 2)
    words = ja.seed_words(77, 16 * T)
    lo, hi = lane_block(rank, world, T)
    r1, r2, _ = run_oracle(O, g, p1, p2, words, n_reads, T, job(), thread_begin=lo, thread_end=hi)
    reads = 2 * (r1.count(b"\n") // 4)
    offsets, totals = exchange_counts(reads, [len(r1), len(r2)])
    for e, data in enumerate((r1, r2)):
        path = os.path.join(tmpdir, "shared_R%d.fq" % (e + 1))
        if rank == 0:
            with open(path, "wb") as fh:
                fh.truncate(totals[1][e])
        dist.barrier()
        mm = np.memmap(path, dtype=np.uint8, mode="r+")
        mm[offsets[e]:offsets[e] + len(data)] = np.frombuffer(data, dtype=np.uint8)
        mm.flush()
    dist.barrier()
    if rank == 0:
        with open(os.path.join(tmpdir, "totals.txt"), "w") as fh:
            fh.write("%d %d %d" % (totals[0], totals[1][0], totals[1][1]))
    dist.destroy_process_group()


def test_two_ranks_assemble_the_single_process_output(O, ja, hs25, tmp_path):
    from helpers import job, run_oracle
    from jackalope_amd.sharding import lane_block
    T, n_reads = 37, 5000
    assert lane_block(0, 2, T) == (0, 19) and lane_block(1, 2, T) == (19, 37) and lane_block(2, 3, 10) == (7, 10)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), T, n_reads), nprocs=2, join=True)
    g = ja.synthetic_genome([60_000, 7_000], seed=3)
    words = ja.seed_words(77, 16 * T)
    w1, w2, _ = run_oracle(O, g, hs25[0], hs25[1], words, n_reads, T, job())
    assert open(tmp_path / "shared_R1.fq", "rb").read() == w1
    assert open(tmp_path / "shared_R2.fq", "rb").read() == w2
    tot = [int(x) for x in open(tmp_path / "totals.txt").read().split()]
    assert tot == [n_reads, len(w1), len(w2)]
