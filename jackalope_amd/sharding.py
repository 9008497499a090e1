"""Multi-GPU sharding of a read-generation job: one process per GPU, lanes split in contiguous blocks,
no data-path collective.  The only exchange is an all-gather of per-rank {reads, bytes_R1, bytes_R2}
(RCCL over xGMI when the backend is "nccl"; gloo works the same for CPU tests), from which every rank
derives its byte offset in the shared FASTQ files and the global totals -- the counterpart of the
reference's single shared output file guarded by `omp critical` (/root/reference/src/hts.h:401-412).
"""
import numpy as np


def lane_block(rank, world, n_lanes):
    """Contiguous lane block [begin, end) of `rank`: as even as possible, first blocks one larger
    (the same rule split_int applies to reads, /root/reference/src/util.h:245-258)."""
    base, extra = divmod(int(n_lanes), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def exchange_counts(reads, bytes_per_end, device=None):
    """All-gather {reads, bytes per read end} over torch.distributed.

    Returns (offsets, totals): offsets[e] = byte offset of this rank's image of read end e in the
    rank-ordered file, totals = (total reads, [total bytes per end]).  Works uninitialised (world = 1)."""
    import torch
    import torch.distributed as dist
    vals = [int(reads)] + [int(b) for b in bytes_per_end]
    if not (dist.is_available() and dist.is_initialized()):
        return [0] * len(bytes_per_end), (vals[0], vals[1:])
    t = torch.tensor(vals, dtype=torch.int64, device=device if device is not None else "cpu")
    gathered = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, t)
    table = np.stack([g.cpu().numpy() for g in gathered])         # [rank][field]
    r = dist.get_rank()
    offsets = [int(table[:r, 1 + e].sum()) for e in range(len(bytes_per_end))]
    totals = (int(table[:, 0].sum()), [int(table[:, 1 + e].sum()) for e in range(len(bytes_per_end))])
    return offsets, totals


def _gather_pairs(a, b, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [(int(a), int(b))]
    t = torch.tensor([int(a), int(b)], dtype=torch.int64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [(int(x[0]), int(x[1])) for x in out]


def open_shard(open_fn, n_lanes, n_units, words_per_lane, device=None):
    """Open this rank's lane shard with set-up work in O(own lanes): the seed-offset exchange of the multi-GPU design.

    A lane's add_n_reads words sit in the seed stream after the n_lanes * 8 words of mt_seeds and after the words of
    all lanes before it; how many words a lane takes depends on its own draws (a haplotype that gets no reads takes
    none), so a rank cannot know its offset without the others.  Every rank therefore opens with the offset that
    holds when every lane before it takes `words_per_lane` (8 for a reference genome, 8 + 16 * n_haplotypes for a
    haplotype set), the ranks all-gather the word ranges their sessions report (jk_session_shard_seed_words: two
    integers per rank, RCCL when the backend is nccl) and a rank whose start is not its predecessor's end opens again
    with the right offset -- in practice never, since a haplotype without reads needs < ~10 reads per lane.

    open_fn(lane_begin, lane_end, seed_offset_words) -> session;  n_units = reads / read ends of the whole run (lanes
    with reads are the first min(n_lanes, n_units)).  Returns the session."""
    import torch.distributed as dist
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    lo, hi = lane_block(rank, world, n_lanes)
    after_mt = 8 * int(n_lanes)
    active = min(int(n_lanes), int(n_units))
    s = open_fn(lo, hi, after_mt + min(lo, active) * int(words_per_lane))
    for _ in range(world + 1):
        b, e = s.shard_seed_words()
        table = _gather_pairs(b, e, device)
        want = [after_mt] + [table[r][1] for r in range(world - 1)]
        if all(table[r][0] == want[r] for r in range(world)):
            return s
        if table[rank][0] != want[rank]:
            s.close()
            s = open_fn(lo, hi, want[rank])
    raise RuntimeError("seed-offset exchange did not settle")
