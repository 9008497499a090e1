#!/usr/bin/env python3
"""Generate the committed golden vectors from the REAL reference code that builds here.

Only the reference's header-only PCG library compiles in this image (oracle/Makefile target `ref`
-> oracle/_ref/libref_pcg.so, built from /root/reference/inst/include/pcg/*.hpp where they lie), so
that is what is captured: pcg64 output streams for fixed sub-seed rows, seeded exactly as
/root/reference/src/pcg.h:48-85 does.  The vectors travel to the GPU box; the reference does not.

Also stored: the known-answer fixtures of the reference's own sequencer tests
(/root/reference/tests/testthat/test-sequencer.R:82-161) and of its mutation-table test
(tests/testthat/test-vcf_IO.R:14-90) as plain data (inputs + expected outputs).

Run from the repo root:  python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def pcg_vectors():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_pcg.so"))
    rows = [
        [0, 0, 0, 0, 0, 0, 0, 0],
        [1, 2, 3, 4, 5, 6, 7, 8],
        [0xFFFFFFFF] * 8,
        [0xDEADBEEF, 0x01234567, 0x89ABCDEF, 0x0BADF00D, 0xCAFEF00D, 0xD15EA5E5, 0x00000001, 0x80000000],
        [3141592653, 589793238, 462643383, 2795028841, 971693993, 751058209, 749445923, 78164062],
    ]
    out = []
    for r in rows:
        w = np.asarray(r, dtype=np.uint32)
        o = np.zeros(64, dtype=np.uint64)
        lib.ref_pcg64_outputs(w.ctypes.data_as(C.c_void_p), C.c_uint64(64), o.ctypes.data_as(C.c_void_p))
        out.append({"sub_seeds": [int(x) for x in r], "outputs": ["%016x" % int(x) for x in o]})
    lib.ref_pcg64_max.restype = C.c_uint64
    return {"source": "pcg64 (setseq_xsl_rr_128_64) of /root/reference/inst/include/pcg/pcg_random.hpp via "
                      "oracle/ref_pcg_driver.cpp; seeding per src/pcg.h:48-85",
            "max": "%016x" % lib.ref_pcg64_max(), "streams": out}


def known_answers():
    # test-sequencer.R:82-87: one quality (255) per nucleotide and position, count 1000
    # test-sequencer.R:91-124 (paired) and :128-161 (mate-pair)
    return {
        "source": "/root/reference/tests/testthat/test-sequencer.R:82-161",
        "chrom": "C" * 25 + "N" * 150 + "T" * 25,
        "read_length": 100, "n_reads": 10000, "frag_len_min": 200, "frag_len_max": 200,
        "profile": {"n_positions": 100, "quals": [255], "cum_counts": [1000]},
        "paired_expected_reads": sorted(["C" * 25 + "N" * 75, "A" * 25 + "N" * 75]),
        "matepair_expected_reads": sorted(["N" * 75 + "T" * 25, "N" * 75 + "G" * 25]),
    }


def pcg_advance_vectors():
    """engine::advance of the reference (pcg_random.hpp:419-434): seed, jump, 8 outputs."""
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_pcg.so"))
    out = []
    for row, steps in [([1, 2, 3, 4, 5, 6, 7, 8], s) for s in (0, 1, 63, 64, 4096, 2 * 2048 * 977, (1 << 40) + 5, (1 << 64) - 1)] + \
                      [([0xDEADBEEF, 0x01234567, 0x89ABCDEF, 0x0BADF00D, 0xCAFEF00D, 0xD15EA5E5, 1, 0x80000000], s)
                       for s in (12345678901, 250_000_000)]:
        w = np.asarray(row, dtype=np.uint32)
        o = np.zeros(8, dtype=np.uint64)
        lib.ref_pcg64_advance_outputs(w.ctypes.data_as(C.c_void_p), C.c_uint64(0), C.c_uint64(steps), C.c_uint64(8),
                                      o.ctypes.data_as(C.c_void_p))
        out.append({"sub_seeds": [int(x) for x in row], "steps": str(steps), "outputs": ["%016x" % int(x) for x in o]})
    return {"source": "pcg64::advance of /root/reference/inst/include/pcg/pcg_random.hpp:419-434 via oracle/ref_pcg_driver.cpp",
            "cases": out}


def vcf_io_fixture():
    """Edits and outcome of /root/reference/tests/testthat/test-vcf_IO.R:14-90 as data (1-based, as the R methods take them)."""
    e1 = [["sub", 1, 6, "T"], ["sub", 2, 6, "A"], ["sub", 3, 6, "A"], ["sub", 3, 7, "T"], ["sub", 4, 6, "G"], ["sub", 4, 8, "T"],
          ["del", 1, 7, 1], ["del", 2, 7, 2], ["del", 4, 9, 1], ["del", 1, 1, 3], ["del", 2, 2, 3], ["del", 3, 1, 3], ["del", 4, 3, 2]]
    e2 = [["del", 1, 9, 1], ["ins", 1, 8, "A"], ["sub", 1, 6, "A"], ["sub", 2, 6, "A"], ["sub", 3, 6, "T"], ["del", 4, 6, 1],
          ["ins", 1, 5, "TT"], ["ins", 2, 5, "TT"], ["ins", 3, 5, "T"], ["ins", 4, 5, "C"], ["sub", 4, 3, "T"],
          ["ins", 2, 2, "AG"], ["del", 3, 2, 2], ["ins", 4, 2, "AG"], ["del", 1, 1, 1]]
    vcf = {"1": [[1, "TCAG", ["G", "T", "TC"], [1, 2, 1, 3]], [6, "CAGT", ["TGT", "AT", "ATGT", "GAT"], [1, 2, 3, 4]]],
           "2": [[1, "TCA", ["CA", "TCAGA", "T", "TCAGT"], [1, 2, 3, 4]], [5, "TC", ["TTTA", "TTT"], [1, 1, 2, 0]],
                 [8, "GT", ["GA"], [1, 0, 0, 0]]]}
    return {"source": "/root/reference/tests/testthat/test-vcf_IO.R:14-90", "chromosome": "TCAGTCAGTC", "n_haps": 4,
            "edits": {"1": e1, "2": e2}, "vcf_rows_pos_ref_alts_genotypes": vcf}


def oracle_digests():
    """sha256 of the FASTQ this repo's CPU oracle writes for a few small jobs.  NOT reference output (the
    reference cannot be built here): a regression anchor that both the oracle and the HIP path must keep hitting."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import oracle_lib as O
    import jackalope_amd as ja
    from golden_jobs import JOBS, run_oracle
    out = {}
    for name in JOBS:
        r1, r2 = run_oracle(ja, O, name)
        out[name] = {"R1_sha256": hashlib.sha256(r1).hexdigest(), "R1_bytes": len(r1), "R1_head": r1[:160].decode(),
                     "R2_sha256": hashlib.sha256(r2).hexdigest() if r2 is not None else None}
    return {"source": "oracle/jk_oracle.cpp (this repo's CPU restatement), jobs defined in tests/golden_jobs.py", "jobs": out}


if __name__ == "__main__":
    with open(os.path.join(HERE, "pcg64_advance_vectors.json"), "w") as fh:
        json.dump(pcg_advance_vectors(), fh, indent=1)
    with open(os.path.join(HERE, "vcf_io_mutations.json"), "w") as fh:
        json.dump(vcf_io_fixture(), fh, indent=1)
    with open(os.path.join(HERE, "oracle_fastq_digests.json"), "w") as fh:
        json.dump(oracle_digests(), fh, indent=1)
    with open(os.path.join(HERE, "pcg64_vectors.json"), "w") as fh:
        json.dump(pcg_vectors(), fh, indent=1)
    with open(os.path.join(HERE, "sequencer_known_answers.json"), "w") as fh:
        json.dump(known_answers(), fh, indent=1)
    print("wrote golden vectors")
