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
