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
    res, desc = fuzz_gpu.pacbio_case(ja, O, rng, case)
    assert res == "ok", desc
