"""A short seeded slice of the randomised parity sweep (tests/fuzz_gpu.py): random Illumina / PacBio jobs, HIP
output byte-identical to the oracle or refused as documented.  Also the two sweep cases that first exposed reads
walking past their window into what earlier reads left in the reference's buffer (duplicates that lost deletions
at a chromosome end; src/hts_pacbio.cpp:277-285, :381-400)."""
import numpy as np
import pytest

import fuzz_gpu

pytestmark = pytest.mark.gpu


def test_seeded_slice(ja, O):
    stats = fuzz_gpu.run(seconds=float("inf"), seed=1, kind="all", max_cases=60, verbose=False)     # bounded by cases, not by time
    assert stats["ok"] + stats["refused"] == 60 and stats["refused"] <= 5


@pytest.mark.parametrize("case", [746, 911])
def test_reads_past_their_window(ja, O, case):
    rng = np.random.default_rng([1, case])
    rng.random()                      # (the draw fuzz_gpu.run makes to pick the kind)
    res, desc = fuzz_gpu.pacbio_case(ja, O, rng, case, round3=False)
    assert res == "ok", desc


@pytest.mark.parametrize("seed,case", [(31, 10), (31, 137), (32, 137)])
def test_round3_sweep_finds(ja, O, seed, case):
    """What the sweep found in the two-kernel PacBio path (`--kind pacbio --seed S --first-case C --cases 1`):
    (31, 10)   130 lanes x 106 reads of up to 28 kb with duplicates: a duplicate that loses deletions patches its mask blocks
               with vector stores; blocks of the next read, written by scalar stores into the same 64-byte line, put the
               old line back (a read's blocks now start on a line of their own);
    (31, 137)  a 1167-base chromosome, reads as long as it: an insertion recorded by pass 1 can be the LAST position of
               append_pool's walk, which then emits L + 1 bases (src/hts_pacbio.cpp:384-388) -- the record is a byte longer;
    (32, 137)  a 5842-base chromosome, one lane: a lane re-walking its masks read a 128-byte line of the vector L1 as an
               earlier read had left it (the scalar stores go past that cache: it is invalidated after the write-back)."""
    rng = np.random.default_rng([seed, case])
    res, desc = fuzz_gpu.pacbio_case(ja, O, rng, case)
    assert res == "ok", desc
