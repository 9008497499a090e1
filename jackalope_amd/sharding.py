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
