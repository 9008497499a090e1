"""GPU: the BASELINE configs at FULL size on one MI355X.  The oracle cannot finish these in seconds, so parity is
(a) byte-equality with the oracle for windows of lanes of the very same job -- every seed and quota derived as in the
full run -- pulled out of the 100-200 GB images with jk_session_fetch_range, and (b) size-independent properties of
the whole output computed on the device (records per read, a record start at every lane start, byte sums of shards).

  configs[1]  100 Mbp, 1 haplotype, 10 M pairs                          test_full_size_config2
  configs[2]  3 Gbp (24 x 125 Mbp), 4 haplotypes, 300 M pairs, 2^21 lanes   test_config3_full_size_four_haplotypes
  configs[3]  3 Gbp x 8 haplotypes, 300 M pairs as 8 lane shards           test_config4_eight_haplotypes_as_eight_shards
  configs[4]  3 Gbp PacBio, 6 M reads of 5-15 kb, 2^20 lanes               test_config5_pacbio_full_size
plus a 4.5 Gbp reference whose last chromosomes lie beyond byte offset 2^32 of the genome buffer."""
import os
import time

import numpy as np
import pytest

from helpers import job, run_oracle, open_hip, DeviceImage

pytestmark = pytest.mark.gpu


def log(msg):
    print("[full-size] " + msg, flush=True)


def lane_offsets(lb):
    return np.concatenate([[0], np.cumsum(lb)]).astype(np.int64)


def check_image_properties(s, lb, n_lanes, n_reads_per_end, seed):
    """4 lines per read; every lane with bytes starts a record ('@') and ends one (newline)."""
    sizes, _ = s.sizes()
    for e in range(len(sizes)):
        img = DeviceImage(s, e)
        off = lane_offsets(lb[e])
        assert int(off[-1]) == sizes[e] == img.n
        assert img.count(10) == 4 * n_reads_per_end
        pick = np.random.default_rng(seed + e).integers(0, n_lanes, size=20000)
        pick = pick[lb[e][pick] > 0]
        assert (img.at(off[pick]) == ord("@")).all()
        assert (img.at(off[pick + 1] - 1) == 10).all()
        del img


def make_reference(ja, n_chroms, chrom_len, seed):
    dev = ja.create_genome(n_chroms, chrom_len, 0, seed_words=ja.seed_words(seed, 8))
    ref = ja.RefGenome([dev.chrom(i) for i in range(n_chroms)], names=dev.names)
    dev.close()
    return ref


def test_full_size_config2(ja, O, hs25):
    n_pairs, T = 10_000_000, 1 << 20
    g = ja.synthetic_genome([100_000_000], seed=2)
    words = ja.seed_words(12345, 16 * T)
    j = job()
    with open_hip(ja, g, (None, None), 150, words, 2 * n_pairs, T, j) as s:
        s.generate()
        sizes, reads = s.sizes()
        assert reads == 2 * n_pairs
        lb = [s.lane_bytes(e, T) for e in range(2)]
        r = [np.frombuffer(s.fetch(e), dtype=np.uint8) for e in range(2)]
        # determinism: a second pass gives the same image
        s.generate()
        assert s.sizes()[0] == sizes
        again = np.frombuffer(s.fetch(0), dtype=np.uint8)
        assert np.array_equal(again, r[0])
        del again
    for e in range(2):
        assert int(lb[e].sum()) == sizes[e] == r[e].size
        # 4 lines per read, one '@' line + one '+' line per read
        assert int(np.count_nonzero(r[e] == 10)) == 4 * n_pairs
        # every lane starts a record; check 2000 random lane starts and the last byte
        off = np.concatenate([[0], np.cumsum(lb[e])[:-1]]).astype(np.int64)
        pick = np.random.default_rng(e).integers(0, T, size=2000)
        assert (r[e][off[pick]] == ord("@")).all() and r[e][-1] == 10
    # oracle parity for windows of lanes of this exact job (all seeds/quotas derived as in the full run)
    for lo, hi in [(0, 24), (524_280, 524_300), (T - 16, T)]:
        o1, o2, _ = run_oracle(O, g, hs25[0], hs25[1], words, 2 * n_pairs, T, j, thread_begin=lo, thread_end=hi)
        for e, o in ((0, o1), (1, o2)):
            a = int(lb[e][:lo].sum())
            n = int(lb[e][lo:hi].sum())
            assert r[e][a:a + n].tobytes() == o, "lanes %d..%d of R%d differ from the oracle" % (lo, hi, e + 1)


def test_lanes_where_a_cut_points_low_word_decided(ja, O, hs25):
    """The alias step's `u < Prob[i]` is decided from the high 32 bits of the exact 64-bit cut point; only a draw whose high
    word EQUALS the cut point's reads the low word (global memory, csrc/jk_illumina_kernel.h `tab_lo`).  That is one alias
    draw in 2^32 -- a 10 M-pair job makes 3.0e9 of them -- so no small case reaches the branch.  The generator notes the
    lanes where it happened; this test runs headline-sized jobs until it has seen the branch at least twice and compares
    exactly those lanes (and their neighbours) with the oracle."""
    n_pairs, T = 10_000_000, 1 << 18
    g = ja.synthetic_genome([100_000_000], seed=2)
    j = job()
    seen = 0
    for rnd in range(8):
        words = ja.seed_words(777 + rnd, 16 * T)
        with open_hip(ja, g, (None, None), 150, words, 2 * n_pairs, T, j) as s:
            s.generate()
            n, lanes = s.rare_branch_lanes()
            assert n == len(lanes) < 40                      # (expected 0.7 per job)
            lb = [s.lane_bytes(e, T) for e in range(2)]
            for lane in sorted(set(lanes)):
                lo, hi = max(lane - 1, 0), min(lane + 2, T)
                o1, o2, _ = run_oracle(O, g, hs25[0], hs25[1], words, 2 * n_pairs, T, j, thread_begin=lo, thread_end=hi)
                for e, o in ((0, o1), (1, o2)):
                    a, nb = int(lb[e][:lo].sum()), int(lb[e][lo:hi].sum())
                    got = s.fetch_range(e, a, nb)
                    assert bytes(got) == o, "lanes %d..%d of R%d (low-word branch in lane %d) differ from the oracle" % (lo, hi, e + 1, lane)
            seen += n
        if seen >= 2:
            break
    assert seen >= 2, "the low-word branch was never taken in %d jobs: is the log wired?" % (rnd + 1)


def test_genome_offsets_beyond_4g(ja, O, hs25):
    """36 chromosomes of 125 Mbp = 4.5 Gbp at one byte per base: the last chromosomes sit beyond byte 2^32 of the
    device genome buffer, so every 64-bit address computation of the read fetch is exercised.  Whole job against the
    oracle (reference-genome path; each lane's quota is split over all 36 chromosomes)."""
    n_chroms, chrom_len = 36, 125_000_000
    t0 = time.time()
    ref = make_reference(ja, n_chroms, chrom_len, seed=9)
    assert sum(ref.sizes()[:35]) > 2 ** 32
    n_pairs, T = 150_000, 3000
    words = ja.seed_words(99, 16 * T)
    j = job()
    with open_hip(ja, ref, (None, None), 150, words, 2 * n_pairs, T, j) as s:
        s.generate()
        sizes, reads = s.sizes()
        r1, r2 = s.fetch(0), s.fetch(1)
        used = s.seed_words_used()
    log("4.5 Gbp reference: generated %d pairs (%.1f s incl. genome)" % (reads // 2, time.time() - t0))
    o1, o2, used_o = run_oracle(O, ref, hs25[0], hs25[1], words, 2 * n_pairs, T, j)
    assert used == used_o
    assert r1 == o1 and r2 == o2
    # reads really came from beyond 2^32: chromosome 35 starts at 35 * 125e6 = 4.375e9
    assert b"@REF-chrom35-" in r1 and b"@REF-chrom34-" in r1


def oracle_hap_windows(O, hs, hs25, words, n_reads, T, windows):
    j = job()
    O.set_windows(windows)
    O.set_chrom_cache(True)
    try:
        return O.illumina_hap(hs, hap_probs=[1.0] * hs.n_haps(), paired=True, n_reads=n_reads, prob_dup=0.02, n_threads=T,
                              read_pool_size=1000, shape=16.0, scale=25.0, fmin=150, fmax=2 ** 32 - 1,
                              prof1=hs25[0], prof2=hs25[1], ins1=j["ins_prob1"], del1=j["del_prob1"],
                              ins2=j["ins_prob2"], del2=j["del_prob2"], words=words)
    finally:
        O.set_windows(None)
        O.set_chrom_cache(False)


def compare_windows(fetch, lb, windows, o1, o2, what):
    """fetch(e, lo_lane, hi_lane) -> bytes of those lanes; the oracle output is the windows' concatenation."""
    at = [0, 0]
    for lo, hi in windows:
        for e, o in ((0, o1), (1, o2)):
            n = int(lb[e][lo:hi].sum())
            got = fetch(e, lo, hi)
            assert len(got) == n
            assert got == o[at[e]:at[e] + n], "%s: lanes %d..%d of R%d differ from the oracle" % (what, lo, hi, e + 1)
            at[e] += n
    assert at[0] == len(o1) and at[1] == len(o2)


def test_config3_full_size_four_haplotypes(ja, O, hs25):
    """BASELINE configs[2]: 3 Gbp (24 x 125 Mbp), 4 haplotypes (14.4 M mutations), 30x PE150 = 300 M pairs on 2^21 lanes,
    198 GB of FASTQ resident in HBM.  (At one byte per base a 3 Gbp genome stays below offset 2^32 of the genome buffer --
    test_genome_offsets_beyond_4g covers that; here it is the image offsets, up to 99 GB per end, that exceed 2^32.)"""
    from jackalope_amd.genome import random_haplotypes_flat
    import torch
    t0 = time.time()
    n_chroms, chrom_len, T = 24, 125_000_000, 1 << 21
    ref = make_reference(ja, n_chroms, chrom_len, seed=3)
    hs = random_haplotypes_flat(ref, 4, seed=31)
    n_pairs = n_chroms * chrom_len * 30 // 300
    words = ja.seed_words(12345, hs.seed_budget(T))
    log("configs[2]: genome + %d mutations + seed words: %.1f s" % (int(hs.n_mut.sum()), time.time() - t0))
    t0 = time.time()
    s = ja.illumina(hs, None, 2 * n_pairs, 150, True, n_threads=T, seed_words=words, _session=True)
    log("open (2^21 lanes x 4 haplotypes x 24 chromosomes): %.2f s" % (time.time() - t0))
    windows = [(0, 3), (T // 2 - 1, T // 2 + 2), (T - 2, T)]
    with s:
        t0 = time.time()
        s.generate()
        sizes, reads = s.sizes()
        log("generate: %d pairs, %.1f + %.1f GB in %.2f s" % (reads // 2, sizes[0] / 1e9, sizes[1] / 1e9, time.time() - t0))
        assert reads == 2 * n_pairs
        assert min(sizes) > 90e9
        used = s.seed_words_used()
        lb = [s.lane_bytes(e, T) for e in range(2)]
        check_image_properties(s, lb, T, n_pairs, seed=5)
        off = [lane_offsets(lb[e]) for e in range(2)]
        assert int(off[0][windows[1][0]]) > 2 ** 32
        got = {(e, lo, hi): s.fetch_range(e, int(off[e][lo]), int(off[e][hi] - off[e][lo])) for e in range(2) for lo, hi in windows}
    torch.cuda.empty_cache()
    t0 = time.time()
    o1, o2, used_o = oracle_hap_windows(O, hs, hs25, words, 2 * n_pairs, T, windows)
    log("oracle windows (sequential planning of 2^21 threads + 96 materialised chromosomes): %.1f s" % (time.time() - t0))
    assert used_o == used
    compare_windows(lambda e, lo, hi: got[(e, lo, hi)], lb, windows, o1, o2, "configs[2]")


def test_config4_eight_haplotypes_as_eight_shards(ja, O, hs25):
    """BASELINE configs[3] on the one GPU there is: 3 Gbp x 8 haplotypes, 300 M pairs, generated (i) unsharded and
    (ii) as the 8 lane shards the 8 ranks of a node would take (lane_begin/lane_end), each shard planned in
    O(own lanes) from the seed-word offset the previous shards report (the seed-offset exchange).  The shards'
    images must be the unsharded image cut at the shard boundaries (sizes, byte sums, position-weighted sums, records),
    and lane windows -- one inside a shard, one across a shard boundary -- must equal the oracle."""
    from jackalope_amd.genome import random_haplotypes_flat
    import torch
    n_chroms, chrom_len, T, n_haps, world = 24, 125_000_000, 1 << 20, 8, 8
    t0 = time.time()
    ref = make_reference(ja, n_chroms, chrom_len, seed=3)
    hs = random_haplotypes_flat(ref, n_haps, seed=31)
    n_pairs = n_chroms * chrom_len * 30 // 300
    words = ja.seed_words(12345, hs.seed_budget(T))
    log("configs[3]: genome + %d mutations: %.1f s" % (int(hs.n_mut.sum()), time.time() - t0))
    per = T // world
    windows = [(5, 7), (4 * per - 1, 4 * per + 1)]
    t0 = time.time()
    with ja.illumina(hs, None, 2 * n_pairs, 150, True, n_threads=T, seed_words=words, _session=True) as s:
        s.generate()
        sizes, reads = s.sizes()
        assert reads == 2 * n_pairs
        used = s.seed_words_used()
        lb = [s.lane_bytes(e, T) for e in range(2)]
        check_image_properties(s, lb, T, n_pairs, seed=6)
        off = [lane_offsets(lb[e]) for e in range(2)]
        whole = {}
        for e in range(2):
            img = DeviceImage(s, e)
            for r in range(world):
                a, b = int(off[e][r * per]), int(off[e][(r + 1) * per])
                whole[(e, r)] = (b - a, img.byte_sum(a, b), img.weighted_sum(a, b), img.count(10, a, b))
            del img
        got = {(e, lo, hi): s.fetch_range(e, int(off[e][lo]), int(off[e][hi] - off[e][lo])) for e in range(2) for lo, hi in windows}
    torch.cuda.empty_cache()
    log("unsharded run + per-shard sums: %.1f s" % (time.time() - t0))
    t0 = time.time()
    seed_at = None
    shard_got = {}
    for r in range(world):
        lo_l, hi_l = r * per, (r + 1) * per
        with ja.illumina(hs, None, 2 * n_pairs, 150, True, n_threads=T, seed_words=words, lane_begin=lo_l, lane_end=hi_l,
                         seed_offset_words=seed_at, _session=True) as s:
            b_w, e_w = s.shard_seed_words()
            if r == 0:
                assert b_w == 8 * T            # right after mt_seeds
            else:
                assert b_w == seed_at
            seed_at = e_w
            s.generate()
            ssz, sreads = s.sizes()
            slb = [s.lane_bytes(e, per) for e in range(2)]
            for e in range(2):
                assert np.array_equal(slb[e], lb[e][lo_l:hi_l])
                img = DeviceImage(s, e)
                assert (img.n, img.byte_sum(), img.weighted_sum(0, img.n), img.count(10)) == whole[(e, r)], "shard %d, R%d" % (r, e + 1)
                del img
            soff = [lane_offsets(slb[e]) for e in range(2)]
            for lo, hi in windows:          # the parts of the oracle windows that fall into this shard
                a, b = max(lo, lo_l), min(hi, hi_l)
                if a < b:
                    for e in range(2):
                        shard_got[(e, a, b)] = s.fetch_range(e, int(soff[e][a - lo_l]), int(soff[e][b - lo_l] - soff[e][a - lo_l]))
        torch.cuda.empty_cache()
    assert seed_at == used                 # the last shard ends where the whole run's seed consumption ends
    log("8 shards: %.1f s" % (time.time() - t0))
    # windows re-assembled from the shards equal the unsharded run's
    for e in range(2):
        for lo, hi in windows:
            parts = b"".join(shard_got[k] for k in sorted(k for k in shard_got if k[0] == e and lo <= k[1] and k[2] <= hi))
            assert parts == got[(e, lo, hi)]
    t0 = time.time()
    o1, o2, used_o = oracle_hap_windows(O, hs, hs25, words, 2 * n_pairs, T, windows)
    log("oracle windows: %.1f s" % (time.time() - t0))
    assert used_o == used
    compare_windows(lambda e, lo, hi: got[(e, lo, hi)], lb, windows, o1, o2, "configs[3]")


def test_config5_pacbio_full_size(ja, O):
    """BASELINE configs[4]: 3 Gbp reference, PacBio reads of 5-15 kb (mean 10 kb), 20x = 6 M reads on 2^20 lanes,
    120 GB of FASTQ, several generator launches."""
    import torch
    n_chroms, chrom_len, T = 24, 125_000_000, 1 << 20
    ref = make_reference(ja, n_chroms, chrom_len, seed=3)
    n_reads = n_chroms * chrom_len * 20 // 10000
    lens = list(range(5000, 15001, 500))
    words = ja.seed_words(4242, 16 * T)
    windows = [(0, 6), (T // 2, T // 2 + 4), (T - 3, T)]
    t0 = time.time()
    with ja.pacbio(ref, None, n_reads, n_threads=T, seed_words=words, custom_read_lengths=lens, _session=True) as s:
        s.generate()
        sizes, reads = s.sizes()
        log("configs[4]: %d reads, %.1f GB, %d launches, %d re-plans: %.1f s" % (reads, sizes[0] / 1e9, s.n_batches(), s.retries(), time.time() - t0))
        assert reads == n_reads and s.n_batches() > 1 and sizes[0] > 100e9
        used = s.seed_words_used()
        lb = [s.lane_bytes(0, T)]
        check_image_properties(s, lb, T, n_reads, seed=7)
        off = lane_offsets(lb[0])
        got = {(lo, hi): s.fetch_range(0, int(off[lo]), int(off[hi] - off[lo])) for lo, hi in windows}
    torch.cuda.empty_cache()
    O.set_windows(windows)
    try:
        o, used_o, tb = O.pacbio_ref(ref, {"custom_read_lengths": lens}, n_reads=n_reads, n_threads=T, words=words)
    finally:
        O.set_windows(None)
    assert used_o == used
    at = 0
    for lo, hi in windows:
        n = int(lb[0][lo:hi].sum())
        assert got[(lo, hi)] == o[at:at + n], "configs[4]: lanes %d..%d differ from the oracle" % (lo, hi)
        at += n
    assert at == len(o)


def test_pacbio_image_guard(ja, O):
    """The PacBio image is sized for the expected read length, the pools for the longest reads; a length model whose
    realised mean is far above its nominal one (min_read_length cutting off most of the log-normal) overruns the
    image.  The compaction must notice before writing past it (JK_KERR_IMAGE_FULL): with re-planning disabled the run
    ends with an error, otherwise it re-plans with a larger image and the result equals the oracle."""
    ref = ja.synthetic_genome([3_000_000, 2_000_000], seed=8)
    n_reads, T = 40_000, 2048
    words = ja.seed_words(11, 16 * T)
    # nominal log-normal mean ~ 8.2 kb; a read below 9 kb is redrawn (up to 10 times), so the realised mean is that of
    # the distribution's upper 38 %, ~12 kb: a third above the 9 kb the image is sized for (+ 12.5 % + 64 MB)
    pb = dict(min_read_length=9000)
    os.environ["JK_PB_NO_IMAGE_RETRY"] = "1"
    try:
        with ja.pacbio(ref, None, n_reads, n_threads=T, seed_words=words, _session=True, **pb) as s:
            with pytest.raises(ja.JackalopeHipError) as e:
                s.generate()
            assert "does not fit" in str(e.value)
    finally:
        del os.environ["JK_PB_NO_IMAGE_RETRY"]
    with ja.pacbio(ref, None, n_reads, n_threads=T, seed_words=words, _session=True, **pb) as s:
        s.generate()
        assert s.retries() >= 1
        sizes, reads = s.sizes()
        assert reads == n_reads
        out = s.fetch(0)
    o, _, _ = O.pacbio_ref(ref, pb, n_reads=n_reads, n_threads=T, words=words)
    assert out == o


def test_scaled_config3_haplotypes(ja, O, hs25):
    """BASELINE configs[2] scaled to one test's budget: 4 chromosomes x 25 Mbp, 4 haplotypes with ~1.2e-3
    mutations/bp, 30x PE150 = 10 M pairs on 2^19 lanes.  Whole-output properties + oracle parity for lane windows
    (the oracle materialises every haplotype chromosome a lane visits, as the reference does)."""
    from jackalope_amd.genome import HapSet
    rng = np.random.default_rng(31)
    ref = ja.synthetic_genome([25_000_000] * 4, seed=3)
    lut = np.frombuffer(b"TCAG", dtype=np.uint8)
    cells = []
    for h in range(4):
        row = []
        for seq in ref.seqs:
            n = seq.size
            pos = np.unique(rng.integers(1, n - 2, size=int(n * 1.2e-3)) & ~np.int64(3))
            kind = rng.choice(3, size=pos.size, p=[1 / 1.2, 0.1 / 1.2, 0.1 / 1.2])
            delta = np.where(kind == 1, 1, np.where(kind == 2, -1, 0))
            shift = np.concatenate([[0], np.cumsum(delta)[:-1]])
            sub, ins, refb = lut[rng.integers(0, 4, size=pos.size)], lut[rng.integers(0, 4, size=pos.size)], seq[pos]
            nuc = [chr(sb) if kd == 0 else (chr(rb) + chr(ib) if kd == 1 else "")
                   for kd, rb, sb, ib in zip(kind.tolist(), refb.tolist(), sub.tolist(), ins.tolist())]
            row.append({"chrom_size": int(n + delta.sum()), "old_pos": pos.tolist(), "new_pos": (pos + shift).tolist(), "nucleos": nuc})
        cells.append(row)
    hs = HapSet(ref, cells)
    n_pairs, T = 10_000_000, 1 << 19
    words = ja.seed_words(777, hs.seed_budget(T))
    s = ja.illumina(hs, None, 2 * n_pairs, 150, True, n_threads=T, seed_words=words, _session=True)
    with s:
        s.generate()
        sizes, reads = s.sizes()
        assert reads == 2 * n_pairs
        lb = [s.lane_bytes(e, T) for e in range(2)]
        r = [np.frombuffer(s.fetch(e), dtype=np.uint8) for e in range(2)]
        used = s.seed_words_used()
    for e in range(2):
        assert int(lb[e].sum()) == sizes[e] == r[e].size
        assert int(np.count_nonzero(r[e] == 10)) == 4 * n_pairs
        off = np.concatenate([[0], np.cumsum(lb[e])[:-1]]).astype(np.int64)
        pick = np.random.default_rng(e).integers(0, T, size=2000)
        assert (r[e][off[pick]] == ord("@")).all()
    j = job()
    for lo, hi in [(0, 2), (T // 2 + 5, T // 2 + 7)]:
        o1, o2, used_o = O.illumina_hap(hs, hap_probs=[1.0] * 4, paired=True, n_reads=2 * n_pairs, prob_dup=0.02, n_threads=T,
                                        read_pool_size=1000, shape=16.0, scale=25.0, fmin=150, fmax=2 ** 32 - 1,
                                        prof1=hs25[0], prof2=hs25[1], ins1=j["ins_prob1"], del1=j["del_prob1"],
                                        ins2=j["ins_prob2"], del2=j["del_prob2"], words=words, thread_begin=lo, thread_end=hi)
        assert used_o == used
        for e, o in ((0, o1), (1, o2)):
            a, n = int(lb[e][:lo].sum()), int(lb[e][lo:hi].sum())
            assert r[e][a:a + n].tobytes() == o, "lanes %d..%d of R%d differ from the oracle" % (lo, hi, e + 1)
