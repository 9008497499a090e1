"""GPU: BASELINE configs[1] at full size (100 Mbp, 30x PE150 = 10 M pairs on one MI355X).  The oracle
cannot finish that in seconds, so parity here is (a) byte-equality with the oracle for a window of
lanes of the very same job, and (b) size-independent properties of the whole output."""
import numpy as np
import pytest

from helpers import job, run_oracle, open_hip

pytestmark = pytest.mark.gpu


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
